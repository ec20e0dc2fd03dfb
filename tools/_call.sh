#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "l14_shapes" 2>&1 | tail -3
