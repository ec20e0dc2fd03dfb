#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 900 python bench.py --steps 300 --sustain-seconds 60 --cpu-seconds 0 > gpurun_out/bench_soak.json 2> gpurun_out/bench_soak.err
python -c "import json; d=json.load(open('gpurun_out/bench_soak.json')); print(d['value'], d['sustained_images_per_sec'], d['sustained'], d['parity']['d_auroc'], d['parity']['max_abs_dscore'])"
