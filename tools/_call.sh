#!/bin/bash
mkdir -p gpurun_out/r4c35
timeout 900 python tools/dual_stream_probe.py 2> gpurun_out/r4c35/dual.err | tee gpurun_out/r4c35/dual_stream_probe.jsonl; tail -5 gpurun_out/r4c35/dual.err
