#!/bin/bash
# scratch driver (round 3, call 17): attention wave-count arms
mkdir -p gpurun_out/r3c17
O=$PWD/gpurun_out/r3c17
timeout 600 python tools/attn_probe.py 30 2>&1 | grep -E "L=197|max\|v-ref" | head -24 | tee $O/attn_probe.txt
