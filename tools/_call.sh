#!/bin/bash
# scratch driver (round 4, call 3): whole GPU suite on the ABI-v2 tree, then the default bench line
mkdir -p gpurun_out/r4c03
O=$PWD/gpurun_out/r4c03
timeout 1800 python -m pytest tests -m gpu -q -x --durations=25 > $O/pytest.txt 2>&1; tail -45 $O/pytest.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -5 $O/bench_default.err
