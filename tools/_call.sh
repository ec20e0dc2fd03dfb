#!/bin/bash
# scratch driver (round 3, call 18): the whole GPU suite on the final tree (timing) + smoke
mkdir -p gpurun_out/r3c18
O=$PWD/gpurun_out/r3c18
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |^real" $O/pytest.txt | head; grep -A13 "slowest" $O/pytest.txt | cut -c1-150
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
