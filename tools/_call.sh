#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
for M in 100864 109056; do for sh in "768 3072 2" "768 768 2" "2304 768 0"; do set -- $sh
  timeout 120 tools/gemm_bench $M $1 $2 $3 20 0 0 3 0x20 2>&1 | grep "BEST"
done; done
