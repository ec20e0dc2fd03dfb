#!/bin/bash
# scratch driver (round 3, call 16): ping-pong kernel with balanced DMA (variant 7) vs shipped (5)
mkdir -p gpurun_out/r3c16
O=$PWD/gpurun_out/r3c16
for sh in "2304 768 0" "3072 768 1" "768 3072 2" "768 768 2"; do set -- $sh
  timeout 200 tools/gemm_bench 100864 $1 $2 $3 20 0 0 3 0xa0 2>&1 | grep -E "BEST|variant 7 vs|variant 5 vs"
done 2>&1 | tee $O/harness.txt
for v in 5 7 5 7; do
  timeout 300 python bench.py --gemm-variant $v --no-drift --cpu-seconds 0 --sustain-seconds 3 > $O/b_$v.json 2> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b_$v.json"))
print("variant $v", round(d["value"]), "img/s", round(d["sustained_images_per_sec"]), "sustained", d["kernel_ms_per_step"]["gemm"], d["sustained"].get("sclk_mhz_mean"), d["sustained"].get("power_w_mean"))
PY
done 2>&1 | tee $O/model.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear" 2>&1 | tail -4 | tee $O/pytest.txt
