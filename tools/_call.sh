#!/bin/bash
# scratch driver (round 3, call 62): the harness-library tests once more on the final harness build
mkdir -p gpurun_out/r3c62
O=$PWD/gpurun_out/r3c62
timeout 110 python -m pytest tests/test_gpu_ln_tail.py tests/test_gpu_qkv_layout.py tests/test_gpu_ln_fold.py tests/test_gpu_kernels.py -m gpu -x -q -k "tail or layout or fold or sliver or tile64 or full_size_variants" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
