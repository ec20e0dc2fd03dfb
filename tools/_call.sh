#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_ingest.py tests/test_gpu_preprocess.py -x -q -m gpu 2>&1 | tail -2
