#!/bin/bash
# scratch driver (round 4, call 15): whole GPU suite on the final tree of this stage
mkdir -p gpurun_out/r4c15
O=$PWD/gpurun_out/r4c15
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.txt 2>&1; tail -14 $O/pytest.txt | cut -c1-200
