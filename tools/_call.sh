#!/bin/bash
# scratch driver (round 3, call 58): kernel time per step with the device left idle between steps (the energy reading)
mkdir -p gpurun_out/r3c58
O=$PWD/gpurun_out/r3c58
for idle in -1 0 5 20 60; do
  timeout 200 python bench.py --no-drift --cpu-seconds 0 --sustain-seconds 0 --steps 48 --warmup 6 --profile-every 1 --idle-ms $idle > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print("idle_ms $idle", d["kernel_ms_per_step"], "sum", round(sum(d["kernel_ms_per_step"].values()),3))
PY
done 2>&1 | tee $O/idle.txt
