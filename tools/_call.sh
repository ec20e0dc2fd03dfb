#!/bin/bash
# scratch driver (round 3, call 38): final tree — whole GPU suite, smoke, default bench
mkdir -p gpurun_out/r3c38
O=$PWD/gpurun_out/r3c38
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |^real" $O/pytest.txt | head -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["sustained"], d["roofline"]["frac"], d["kernel_ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["value_hoisted"], d["parity"]["meets_1e-4"], d["parity"]["d_auroc"], d["parity"]["d_fpr95"])
PY
