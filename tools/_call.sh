#!/bin/bash
# scratch driver (round 3, call 25): bench line with the HBM-kernel rooflines; the exact-fp32 arm's own line
mkdir -p gpurun_out/r3c25
O=$PWD/gpurun_out/r3c25
timeout 600 python bench.py --no-drift --cpu-seconds 0 > $O/bench_quick.json 2> $O/b.err; tail -2 $O/b.err
timeout 600 python bench.py --precision fp32 --no-drift --cpu-seconds 0 --steps 5 --warmup 2 --sustain-seconds 0 > $O/bench_fp32arm.json 2>> $O/b.err
python - <<PY
import json
d=json.load(open("$O/bench_quick.json"))
print(round(d["value"]), d["roofline_hbm_kernels"])
d=json.load(open("$O/bench_fp32arm.json"))
print("fp32 arm", round(d["value"]), d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["kernel_ms_per_step"])
PY
