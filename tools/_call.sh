#!/bin/bash
# scratch driver (round 3, call 12): energy per component of the ping-pong GEMM (ablation bits x power / clock sampling)
mkdir -p gpurun_out/r3c12
O=$PWD/gpurun_out/r3c12
for sh in "2304 768 0 qkv" "3072 768 1 fc1" "768 3072 2 fc2"; do set -- $sh
  for dbg in 0 4 16 5 6 7; do
    bash tools/smi_sample.sh $4_$dbg tools/gemm_bench 100864 $1 $2 $3 3000 0 $dbg 3 0x20 2>&1 | grep -E "BEST|busy samples" | tr '\n' ' '; echo
  done
done 2>&1 | tee $O/energy.txt
