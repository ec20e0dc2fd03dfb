#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
bash tools/profile.sh r02_d > gpurun_out/profile.log 2>&1
timeout 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 600 python bench.py --ckpt ViT-L/14 --batch 256 --no-drift --cpu-seconds 0 > gpurun_out/bench_L14_fp16.json 2>/dev/null
cat gpurun_out/pytest.log; tail -2 gpurun_out/smoke.log; for f in default L14_fp16; do python -c "import json; d=json.load(open('gpurun_out/bench_$f.json')); print('$f', d['value'], d['sustained_images_per_sec'], d['kernel_ms_per_step'], d['roofline']['frac'])"; done
