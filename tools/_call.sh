#!/bin/bash
# scratch driver for one gpurun call (round 3, call 1: measurements that steer the round)
mkdir -p gpurun_out/r3c1
O=$PWD/gpurun_out/r3c1
R=$PWD
(timeout 300 python tools/blas_yardstick.py > $O/blas.txt 2>&1)
echo "== blas"; cat $O/blas.txt | tail -10
(timeout 900 python tools/hf_gpu_probe.py ViT-B/16 2048 1024 512 fp16-exact fp16,bf16 > $O/hf_probe.json 2> $O/hf_probe.err)
echo "== hf probe"; tail -c 3000 $O/hf_probe.json; tail -5 $O/hf_probe.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "TCC_[A-Z0-9_]*" $O/counters.txt | sort -u | tr '\n' ' ' | head -c 3000; echo
for sh in "2304 768 0 qkv" "3072 768 1 fc1"; do set -- $sh
  for pass in "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE TCC_TAG_STALL_sum"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$4_$tag -o p -- $R/tools/gemm_bench 100864 $1 $2 $3 2 0 0 3 0x28 > $O/pmc_$4_$tag.log 2>&1
    db=$(find $O/pmc_$4_$tag \( -name "*_results.db" -o -name "*counter_collection.csv" \) | head -1)
    echo "== $4 $pass"; [ -n "$db" ] && python $R/tools/pmc_summary.py $db | tee $O/pmc_$4_$tag.txt
  done
done
cd $R
for sh in "2304 768 0" "3072 768 1" "768 3072 2" "768 768 2"; do set -- $sh
  timeout 120 tools/gemm_bench 100864 $1 $2 $3 20 0 0 3 0x28 2>&1 | grep "BEST"
done
rm -rf $O/pmc_*/ 2>/dev/null; find $O -name "*.db" -size +20M -delete
