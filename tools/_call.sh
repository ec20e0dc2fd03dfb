#!/bin/bash
# scratch driver (round 3, call 57): bench.py --gpus 2 on the one GPU of the box (two ranks share it: gloo fallback) — the N > 1 path of the final tree
mkdir -p gpurun_out/r3c57
O=$PWD/gpurun_out/r3c57
timeout 240 python bench.py --gpus 2 --steps 6 --warmup 2 --sustain-seconds 0 > $O/bench2.json 2> $O/bench2.err; echo rc=$?; tail -2 $O/bench2.err
python - <<PY
import json
d=json.load(open("$O/bench2.json"))
print(d["n_gpus"], round(d["value"]), d["ms_per_step"], d.get("collective"), d["config"]["parallelism"])
PY
