#!/bin/bash
# scratch driver (round 3, call 24): final tree — whole GPU suite, default bench, rocprofv3 passes
mkdir -p gpurun_out/r3c24
O=$PWD/gpurun_out/r3c24
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |^real" $O/pytest.txt | head -5
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["sustained"], d["roofline"]["frac"], d["kernel_ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["value_hoisted"], d["parity"]["meets_1e-4"])
PY
bash tools/profile.sh r03_i > $O/profile.log 2>&1; tail -3 $O/profile.log
