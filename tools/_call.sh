#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "linear" 2>&1 | tail -3 > gpurun_out/pytest.log
cat gpurun_out/pytest.log
timeout 300 python tools/mlp_probe.py 12 fp16 3 5 > gpurun_out/mlp_probe.txt 2>&1
cat gpurun_out/mlp_probe.txt
for v in 3 5; do python bench.py --no-drift --cpu-seconds 0 --gemm-variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($v, d['value'], d['sustained_images_per_sec'], d['kernel_ms_per_step'])"; done
