#!/bin/bash
# scratch driver (round 3, call 20): two-pass attention arms
mkdir -p gpurun_out/r3c20
O=$PWD/gpurun_out/r3c20
timeout 600 python tools/attn_probe.py 30 2>&1 | grep -E "L=197|max\|v-ref" | head -14 | tee $O/attn_probe.txt
