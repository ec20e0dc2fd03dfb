#!/bin/bash
# sample sclk/power while a command runs
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/smi.log 2>&1 &
SP=$!
python bench.py --steps 200 --warmup 3 --cpu-seconds 0 --no-profile 2>/dev/null | tail -1 | cut -c1-160
kill $SP 2>/dev/null
sort gpurun_out/smi.log | uniq -c | sort -rn | head -12
