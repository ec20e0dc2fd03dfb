#!/bin/bash
# scratch driver (round 3, call 14): X panels streamed with the nt hint in the ping-pong kernel (dbg bit 32), harness + model + traffic
mkdir -p gpurun_out/r3c14
O=$PWD/gpurun_out/r3c14
R=$PWD
for sh in "2304 768 0" "3072 768 1" "768 3072 2"; do set -- $sh
  for dbg in 0 32 0 32; do
    echo -n "dbg $dbg: "; timeout 200 tools/gemm_bench 100864 $1 $2 $3 20 0 $dbg 3 0x20 2>&1 | grep -E "BEST"
  done
done 2>&1 | tee $O/harness.txt
for i in 1 2; do for dbg in 0 32; do
  timeout 300 python bench.py --gemm-variant 5 --gemm-dbg $dbg --no-drift --cpu-seconds 0 --sustain-seconds 3 > $O/b_$dbg.$i.json 2> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b_$dbg.$i.json"))
print("gemm-dbg $dbg", round(d["value"]), "img/s", round(d["sustained_images_per_sec"]), "sustained", d["kernel_ms_per_step"]["gemm"], d["sustained"].get("sclk_mhz_mean"), d["sustained"].get("power_w_mean"))
PY
done; done 2>&1 | tee $O/model_ab.txt
cd /tmp && export TMPDIR=/tmp
for dbg in 0 32; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $O/f_$dbg -o f -- $R/tools/gemm_bench 100864 2304 768 0 2 0 $dbg 3 0x20 > $O/f_$dbg.log 2>&1
  db=$(find $O/f_$dbg -name "*_results.db" | head -1); echo "== QKV harness dbg $dbg"; python $R/tools/pmc_summary.py $db | grep -E "kernel|gemm_pp"
done 2>&1 | tee $O/traffic.txt
rm -rf $O/f_0 $O/f_32
