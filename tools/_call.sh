#!/bin/bash
# scratch driver (round 3, call 9): rocprofv3 passes of the final tree, other checkpoints, 2-rank logic check
mkdir -p gpurun_out/r3c9
O=$PWD/gpurun_out/r3c9
bash tools/profile.sh r03_e > $O/profile.log 2>&1; tail -12 $O/profile.log
for cfg in "ViT-L/14 fp16 256 L14_fp16" "ViT-B/32 fp16 512 B32_fp16" "ViT-B/16 bf16 512 B16_bf16"; do set -- $cfg
  timeout 600 python bench.py --ckpt $1 --precision $2 --batch $3 --no-drift --cpu-seconds 0 > $O/bench_$4.json 2>> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/bench_$4.json"))
print("$4", round(d["value"]), "img/s", round(d["ms_per_step"],2), "ms", round(d["sustained_images_per_sec"]), "sustained", d["kernel_ms_per_step"], round(d["roofline"]["achieved"]), d["sustained"].get("sclk_mhz_mean"), d["sustained"].get("power_w_mean"))
PY
done 2>&1 | tee $O/others.txt
timeout 600 python bench.py --gpus 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 > $O/bench_2ranks.json 2> $O/b2.err; tail -c 600 $O/bench_2ranks.json; tail -3 $O/b2.err
