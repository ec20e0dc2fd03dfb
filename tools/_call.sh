#!/bin/bash
mkdir -p gpurun_out/r4c59
timeout 1500 python tools/jpeg_sweep.py --n 3000 2>/tmp/s.err | tail -1 | tee gpurun_out/r4c59/jpeg_sweep.json | cut -c1-1500; grep -i "Traceback\|Error" -A3 /tmp/s.err | head -8
