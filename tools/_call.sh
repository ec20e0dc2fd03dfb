#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 60 tools/valu_probe > gpurun_out/valu_probe.txt 2>&1; cat gpurun_out/valu_probe.txt
