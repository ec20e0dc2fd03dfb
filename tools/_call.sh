#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 600 python bench.py --ckpt ViT-L/14 --batch 256 --no-drift --cpu-seconds 0 > gpurun_out/bench_L14_fp16.json 2>/dev/null
timeout 600 python bench.py --ckpt ViT-B/32 --no-drift --cpu-seconds 0 > gpurun_out/bench_B32_fp16.json 2>/dev/null
timeout 600 python bench.py --precision bf16 --no-drift --cpu-seconds 0 > gpurun_out/bench_B16_bf16.json 2>/dev/null
timeout 600 python bench.py --ckpt ViT-L/14 --batch 256 --no-drift --cpu-seconds 0 --gemm-variant 3 > gpurun_out/bench_L14_fp16_v3.json 2>/dev/null
for f in L14_fp16 B32_fp16 B16_bf16 L14_fp16_v3; do python -c "import json; d=json.load(open('gpurun_out/bench_$f.json')); print('$f', round(d['value']), round(d['sustained_images_per_sec']), d['kernel_ms_per_step'], round(d['roofline']['achieved']))"; done
