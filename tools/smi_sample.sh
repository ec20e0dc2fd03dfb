#!/bin/bash
# smi_sample.sh <label> <command...>: sample sclk / package power every 250 ms while the command runs
label=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/smi_$label.log 2>&1 &
SP=$!
"$@"
kill $SP 2>/dev/null
awk '$2>1000 && $4>300 {n++; c+=$2; w+=$4} END{if(n) printf "%s: %d busy samples, sclk %.0f MHz, power %.0f W\n", "'$label'", n, c/n, w/n; else print "'$label': no busy samples"}' gpurun_out/smi_$label.log
