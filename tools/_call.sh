#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 300 python tools/attn_probe.py 20 2>&1 | grep -v amdgpu.ids | grep "L=197"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3
for v in 1 2; do python - <<PY
import json,subprocess,sys
PY
done
python bench.py --no-drift --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['sustained_images_per_sec'], d['kernel_ms_per_step'])"
