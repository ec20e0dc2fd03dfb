#!/bin/bash
# scratch driver (round 3, call 26): GEMM row padding into the workspace — model tests, ragged batches, L/14 at batch 255 vs 256
mkdir -p gpurun_out/r3c26
O=$PWD/gpurun_out/r3c26
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_round2.py tests/test_gpu_configs.py tests/test_gpu_maha.py tests/test_gpu_metrics.py -m gpu -x -q 2>&1 | tail -3
for cfg in "ViT-L/14 256" "ViT-L/14 255" "ViT-L/14 256" "ViT-L/14 255" "ViT-B/16 512" "ViT-B/16 500" "ViT-B/16 300"; do set -- $cfg
  timeout 300 python bench.py --ckpt $1 --batch $2 --no-drift --cpu-seconds 0 --sustain-seconds 3 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print("$1 batch $2:", round(d["value"]), "img/s", round(d["sustained_images_per_sec"]), "sustained", round(d["ms_per_step"],2), "ms", d["kernel_ms_per_step"]["gemm"], round(d["roofline"]["achieved"]))
PY
done 2>&1 | tee $O/bench.txt
