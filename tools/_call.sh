#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 300 python tools/attn_probe.py 20 2>&1 | grep -v amdgpu.ids | head -12
