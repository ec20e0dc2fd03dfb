#!/bin/bash
# scratch driver (round 3, call 22): pre_layrnorm + layer_norm1 fused — model tests, smoke, bench A/B vs previous numbers
mkdir -p gpurun_out/r3c22
O=$PWD/gpurun_out/r3c22
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_round2.py "tests/test_gpu_configs.py::test_config1_imagenet10_vs_imagenet20_b16_batch64" -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --no-drift --cpu-seconds 0 --sustain-seconds 3 > $O/b.$i.json 2> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.$i.json"))
print(round(d["value"]), "img/s", round(d["sustained_images_per_sec"]), "sustained", d["kernel_ms_per_step"])
PY
done 2>&1 | tee $O/bench.txt
