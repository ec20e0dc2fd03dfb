#!/bin/bash
# scratch driver (round 4, call 40): shared decode pool / packed pipe — tests, config 3 end to end from JPEG files
mkdir -p gpurun_out/r4c40
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_configs.py tests/test_refine.py -x -q -m gpu 2>&1 | tail -4
timeout 1500 python tools/e2e_jpeg_config3.py > gpurun_out/r4c40/e2e_jpeg_config3.json 2> gpurun_out/r4c40/e2e.err; grep -i "error\|Traceback" gpurun_out/r4c40/e2e.err | head -3; cat gpurun_out/r4c40/e2e_jpeg_config3.json | cut -c1-1200
