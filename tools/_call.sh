#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
bash tools/profile.sh r02_c > gpurun_out/profile.log 2>&1
timeout 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/pytest.log; tail -2 gpurun_out/smoke.log; tail -8 gpurun_out/profile.log; cat gpurun_out/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['sustained_images_per_sec'], d['kernel_ms_per_step'], d['roofline']['frac'], d['parity']['d_auroc'])"
