#!/bin/bash
# scratch driver (round 3, call 19): the default bench line of the final tree + a 60-second soak
mkdir -p gpurun_out/r3c19
O=$PWD/gpurun_out/r3c19
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
( time timeout 600 python bench.py --steps 300 --warmup 5 --sustain-seconds 60 --no-drift --cpu-seconds 0 ) > $O/soak.json 2> $O/soak.err; tail -3 $O/soak.err
python - <<PY
import json
for f in ("bench","soak"):
    d=json.load(open("$O/%s.json"%f))
    print(f, round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["sustained"], d["roofline"]["frac"], d["kernel_ms_per_step"])
PY
