#!/bin/bash
# scratch driver (round 4, call 33): decode processes sized by the CPU quota; CPU baseline on the quota's threads; default bench
mkdir -p gpurun_out/r4c33
O=$PWD/gpurun_out/r4c33
timeout 900 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_ingest.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ingest", {k: round(v["images_per_sec"]) for k, v in d["ingest"].items()}, d.get("leg_seconds"), "frac", d["roofline"]["frac"])
c=d["cpu_baseline"]; print("cpu", c["value"], c["value_hoisted"], c["cores"], c.get("cpu_quota_cores"), c["seconds"], c.get("parity_max_abs_dscore_vs_native"))
print(d["ingest"]["host_jpeg"])
PY
