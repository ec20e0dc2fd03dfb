#!/bin/bash
mkdir -p gpurun_out/r4c56
O=$PWD/gpurun_out/r4c56
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -2
timeout 1500 python tools/e2e_jpeg_config3.py > $O/e2e_jpeg_config3.json 2> $O/e2e.err; grep -i "error\|Traceback" $O/e2e.err | head -3; cut -c1-200 $O/e2e_jpeg_config3.json
MCM_GPU_JPEG=0 timeout 1500 python tools/e2e_jpeg_config3.py > $O/e2e_jpeg_config3_pillow.json 2> $O/e2e_p.err; cut -c1-200 $O/e2e_jpeg_config3_pillow.json
