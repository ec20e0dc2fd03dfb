#!/bin/bash
# scratch driver for one gpurun call (round 3, call 3: full GPU suite + default bench with the new legs)
mkdir -p gpurun_out/r3c3
O=$PWD/gpurun_out/r3c3
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.txt 2>&1
tail -40 $O/pytest.txt
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 6000 $O/bench.json; tail -5 $O/bench.err
