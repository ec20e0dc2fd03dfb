#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "linear" 2>&1 | tail -3
for sh in "2304 768 0" "3072 768 1" "768 3072 2"; do
  set -- $sh
  timeout 120 tools/gemm_bench 100864 $1 $2 $3 20 0 0 3 0x28 2>&1 | grep "BEST"
done
# a batch that is not a multiple of 256: 384 images -> M = 75648
for sh in "2304 768 0" "3072 768 1"; do
  set -- $sh
  timeout 120 tools/gemm_bench 75648 $1 $2 $3 20 0 0 3 0x28 2>&1 | grep "BEST\|differing"
done
