#!/bin/bash
# scratch driver (round 3, call 13): C-ABI example program, config-5 CLI tests
mkdir -p gpurun_out/r3c13
O=$PWD/gpurun_out/r3c13
( time timeout 2400 python -m pytest tests/test_gpu_c_abi.py tests/test_gpu_configs.py -m gpu -q --durations=8 -s ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |abi_example|scores\[0" $O/pytest.txt | cut -c1-300 | head -30
