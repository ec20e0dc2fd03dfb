#!/bin/bash
# scratch driver (round 3, call 52): final tree — whole GPU suite and smoke
mkdir -p gpurun_out/r3c52
O=$PWD/gpurun_out/r3c52
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |^real" $O/pytest.txt | head -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
