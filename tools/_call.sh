#!/bin/bash
# scratch driver (round 3, call 55): new kernel-choice rule — the tests whose shapes cross it
mkdir -p gpurun_out/r3c55
O=$PWD/gpurun_out/r3c55
( time timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_round2.py tests/test_gpu_configs.py tests/test_gpu_ln_fold.py tests/test_gpu_ln_tail.py tests/test_gpu_qkv_layout.py tests/test_gpu_c_abi.py -m gpu -x -q ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |^real" $O/pytest.txt | head -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
