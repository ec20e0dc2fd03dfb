#!/bin/bash
# scratch driver (round 3, call 37): residual rows requested two chunks ahead (three buffers; spills 4 again) vs one ahead
mkdir -p gpurun_out/r3c37
O=$PWD/gpurun_out/r3c37
for rep in 1 2; do for b in gemm_bench_old gemm_bench; do for shp in "768 3072 2" "768 768 2"; do
  echo -n "$b " >> $O/gemm.txt; timeout 300 tools/$b 100864 $shp 1500 0 0 3 0x20 2>&1 | grep -E "BEST" >> $O/gemm.txt
done; done; done
cat $O/gemm.txt | cut -c1-120
one() {
  timeout 600 python tools/bench_with_lib.py mcm_amd/$2 --no-drift --cpu-seconds 0 --steps 40 > $O/b_$1.json 2> $O/b_$1.err || tail -3 $O/b_$1.err
  python - <<PY
import json
d=json.load(open("$O/b_$1.json"))
print("$1", round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["kernel_ms_per_step"]["gemm"], d["sustained"].get("sclk_mhz_mean"))
PY
}
for rep in 1 2 3; do one old_$rep libmcm_hip_old.so; one new_$rep libmcm_hip.so; done 2>&1 | tee $O/bench.txt
