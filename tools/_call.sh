#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
for i in 1 2; do python bench.py --no-drift --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['sustained_images_per_sec'], d['kernel_ms_per_step'])"; done
git stash -q 2>/dev/null
