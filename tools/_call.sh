#!/bin/bash
# scratch driver (round 4, call 26): streaming-store pack + full default bench line
mkdir -p gpurun_out/r4c26
O=$PWD/gpurun_out/r4c26
timeout 600 python tools/ingest_probe.py > $O/ingest_probe.jsonl 2> $O/ingest_probe.err; grep -v 'h2d\|workload' $O/ingest_probe.jsonl; tail -3 $O/ingest_probe.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ingest", {k: round(v["images_per_sec"]) for k, v in d["ingest"].items()}, d.get("leg_seconds"), "frac", d["roofline"]["frac"])
PY
