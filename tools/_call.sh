#!/bin/bash
# scratch driver (round 4, call 6): attention q-block deal rotated per workgroup (harness arm 11) A/B; attention + config tests
mkdir -p gpurun_out/r4c06
O=$PWD/gpurun_out/r4c06
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -k "attention or config2" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
B="--no-drift --cpu-seconds 0 --ingest none --no-arms --sustain-seconds 3"
for v in 1 11 1 11; do
  timeout 300 python bench.py $B --attn-variant $v > $O/bench_attn$v.json 2> $O/bench_attn$v.err
  python - <<PY
import json
d=json.loads(open("$O/bench_attn$v.json").read().strip().splitlines()[-1])
print("attn variant $v", round(d["value"]), d["kernel_ms_per_step"], d.get("sustained",{}).get("sclk_mhz_mean"), d.get("sustained",{}).get("power_w_mean"))
PY
done
for v in 1 11; do timeout 300 python bench.py $B --attn-variant $v --ckpt ViT-L/14 --batch 256 > $O/bench_L14_attn$v.json 2> $O/bench_L14_attn$v.err
  python - <<PY
import json
d=json.loads(open("$O/bench_L14_attn$v.json").read().strip().splitlines()[-1])
print("L/14 attn variant $v", round(d["value"]), d["kernel_ms_per_step"])
PY
done
for v in 1 11; do timeout 300 python bench.py $B --attn-variant $v --ckpt ViT-B/32 > $O/bench_B32_attn$v.json 2> $O/bench_B32_attn$v.err
  python - <<PY
import json
d=json.loads(open("$O/bench_B32_attn$v.json").read().strip().splitlines()[-1])
print("B/32 attn variant $v", round(d["value"]), d["kernel_ms_per_step"])
PY
done
