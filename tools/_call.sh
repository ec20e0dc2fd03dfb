#!/bin/bash
# scratch driver (round 3, call 10): the whole GPU suite on the final tree + smoke
mkdir -p gpurun_out/r3c10
O=$PWD/gpurun_out/r3c10
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |slowest" -A13 $O/pytest.txt | tail -22
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
