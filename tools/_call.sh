#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
for rep in 1 2; do
for sh in "2304 768 0" "3072 768 1" "768 3072 2" "768 768 2"; do
  set -- $sh
  echo -n "v2  "; timeout 120 tools/gemm_bench_pfv2 100864 $1 $2 $3 20 0 0 3 0x28 2>&1 | grep "BEST"
  echo -n "v2a "; timeout 120 tools/gemm_bench 100864 $1 $2 $3 20 0 0 3 0x28 2>&1 | grep "BEST\|differing elem" | grep -v "variant 3" | tr '\n' ' '; echo
done; done > gpurun_out/pp.txt 2>&1
cat gpurun_out/pp.txt
