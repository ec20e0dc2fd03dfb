#!/bin/bash
mkdir -p gpurun_out/r4c61
timeout 1500 python tools/resize_sweep.py --n 720 2>/tmp/s.err | tail -1 | tee gpurun_out/r4c61/resize_sweep.json | cut -c1-900; grep -i "Traceback\|Error" -A3 /tmp/s.err | head -8
