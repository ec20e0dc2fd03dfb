#!/bin/bash
# scratch driver (round 3, call 15): harness fidelity after moving the in-loop ablation bits out; product bench
mkdir -p gpurun_out/r3c15
O=$PWD/gpurun_out/r3c15
for sh in "2304 768 0" "3072 768 1" "768 3072 2" "768 768 2"; do set -- $sh
  timeout 200 tools/gemm_bench 100864 $1 $2 $3 20 0 0 3 0x28 2>&1 | grep -E "BEST"
done 2>&1 | tee $O/harness.txt
for v in 5 -1 5 -1; do
  timeout 300 python bench.py --gemm-variant $v --no-drift --cpu-seconds 0 --sustain-seconds 3 > $O/b_$v.json 2> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b_$v.json"))
print("variant $v", round(d["value"]), "img/s", round(d["sustained_images_per_sec"]), "sustained", d["kernel_ms_per_step"]["gemm"], d["sustained"].get("sclk_mhz_mean"), d["sustained"].get("power_w_mean"))
PY
done 2>&1 | tee $O/model.txt
