#!/bin/bash
# scratch driver for one `gpurun` call (rewritten per call during development; the last one ran the JPEG / ingest / preprocess
# GPU tests).  Kept so that `gpurun -- 'bash tools/_call.sh'` is always a valid smoke of the ingest path.
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_ingest.py tests/test_gpu_preprocess.py -x -q -m gpu 2>&1 | tail -2
