#!/bin/bash
mkdir -p gpurun_out/r4c48
O=$PWD/gpurun_out/r4c48
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python tools/e2e_jpeg_config3.py > $O/e2e_jpeg_config3.json 2> $O/e2e.err; grep -i "error\|Traceback" $O/e2e.err | head -3; cut -c1-330 $O/e2e_jpeg_config3.json; grep -o '"refine_rescored[^}]*}' $O/e2e_jpeg_config3.json
