#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 12 --no-drift --cpu-seconds 0 --no-arms --no-live-traffic --sustain-seconds 0 --ingest host-jpeg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k: (round(v.get('images_per_sec',-1)), v.get('pipe_seconds_per_batch'), v.get('error')) for k,v in d['ingest'].items()})"
