#!/bin/bash
# scratch driver (round 3, call 47): LayerNorm in the tail of the residual GEMMs — correctness first, then A/B vs the
# library built with -DMCM_NO_LN_TAIL (every LayerNorm launched)
mkdir -p gpurun_out/r3c47
O=$PWD/gpurun_out/r3c47
timeout 900 python -m pytest tests/test_gpu_ln_tail.py -m gpu -x -q > $O/pytest_tail.txt 2>&1; tail -4 $O/pytest_tail.txt
if grep -q "failed\|error" $O/pytest_tail.txt; then grep -E "^E " $O/pytest_tail.txt | head -20; fi
one() {
  timeout 600 python tools/bench_with_lib.py mcm_amd/$2 --no-drift --cpu-seconds 0 --steps 40 > $O/b_$1.json 2> $O/b_$1.err || tail -3 $O/b_$1.err
  python - <<PY
import json
d=json.load(open("$O/b_$1.json"))
print("$1", round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["kernel_ms_per_step"], d["sustained"].get("sclk_mhz_mean"))
PY
}
for rep in 1 2 3; do one launches_$rep libmcm_hip_notail.so; one tail_$rep libmcm_hip.so; done 2>&1 | tee $O/bench.txt
