#!/bin/bash
# scratch driver (round 4, call 42): JPEG device route in the loader — tests, host_jpeg leg both routes, e2e config 3
mkdir -p gpurun_out/r4c42
O=$PWD/gpurun_out/r4c42
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -5
for route in 1 0; do
MCM_GPU_JPEG=$route timeout 600 python bench.py --steps 12 --no-drift --cpu-seconds 0 --no-arms --no-live-traffic --sustain-seconds 0 --ingest host-jpeg > $O/bench_jpeg_route$route.json 2> $O/bench_jpeg_route$route.err
python - <<PY
import json
d=json.loads(open("$O/bench_jpeg_route$route.json").read().strip().splitlines()[-1])
print("route", $route, "value", round(d["value"]), {k: (round(v.get("images_per_sec", -1)), v.get("decoder", v.get("error"))) for k, v in d["ingest"].items()}, d.get("leg_seconds"))
PY
done
timeout 1500 python tools/e2e_jpeg_config3.py > $O/e2e_jpeg_config3.json 2> $O/e2e.err; grep -i "error\|Traceback" $O/e2e.err | head -3; cut -c1-400 $O/e2e_jpeg_config3.json
