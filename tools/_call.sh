#!/bin/bash
mkdir -p gpurun_out/r4c41
timeout 900 python -m pytest tests/test_gpu_jpeg.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r4c41/pytest.txt
