#!/bin/bash
# scratch driver for gpurun calls (rewritten per call; results land under gpurun_out/)
mkdir -p gpurun_out/scratch
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/scratch/pytest.txt 2>&1; tail -3 gpurun_out/scratch/pytest.txt
