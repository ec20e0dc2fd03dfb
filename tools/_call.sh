#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks.err
wc -l gpurun_out/bench_2ranks.json; python -c "import json; d=json.load(open('gpurun_out/bench_2ranks.json')); print(d['n_gpus'], d['value'], d.get('collective'), d['config']['parallelism'])"
