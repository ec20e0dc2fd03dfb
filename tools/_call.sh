#!/bin/bash
# scratch driver (round 3, call 59): timed region after 3 / 8 / 16 warm-up steps (clock ramp at the start of a run)
mkdir -p gpurun_out/r3c59
O=$PWD/gpurun_out/r3c59
for w in 3 16 8 3 16; do
  timeout 200 python bench.py --no-drift --cpu-seconds 0 --sustain-seconds 0 --steps 20 --warmup $w > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print("warmup $w", round(d["value"]), round(d["ms_per_step"],4), d["kernel_ms_per_step"]["gemm"])
PY
done 2>&1 | tee $O/warm.txt
