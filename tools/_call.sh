#!/bin/bash
mkdir -p gpurun_out/r4c34
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r4c34/pytest.txt
