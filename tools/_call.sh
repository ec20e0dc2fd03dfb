#!/bin/bash
mkdir -p gpurun_out/r4c54
O=$PWD/gpurun_out/r4c54
for w in 12 14 15 16 20; do
MCM_DECODE_WORKERS=$w timeout 600 python bench.py --steps 12 --no-drift --cpu-seconds 0 --no-arms --no-live-traffic --sustain-seconds 0 --ingest host-jpeg > $O/bench_jpeg_$w.json 2> $O/bench_jpeg_$w.err
python - <<PY
import json
d=json.loads(open("$O/bench_jpeg_$w.json").read().strip().splitlines()[-1])
print($w, {k: (round(v.get("images_per_sec", -1)), v.get("pipe_seconds_per_batch"), v.get("error")) for k, v in d["ingest"].items()})
PY
done
