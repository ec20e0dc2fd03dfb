#!/bin/bash
# scratch driver (round 4, call 47): whole GPU suite + smoke + default bench on the tree with the JPEG device route
mkdir -p gpurun_out/r4c47
O=$PWD/gpurun_out/r4c47
timeout 1800 python -m pytest tests -q -m gpu --durations=6 > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ingest", {k: round(v.get("images_per_sec", -1)) for k, v in d["ingest"].items()}, d.get("leg_seconds"), "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
c=d["cpu_baseline"]; print("cpu", c.get("value"), c.get("cores"))
print("meets", d["parity"].get("meets_1e-4"), "arms", {k: round(v.get("images_per_sec", -1)) for k, v in d["arms"].items()})
PY
