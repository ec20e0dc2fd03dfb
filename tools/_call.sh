#!/bin/bash
# scratch driver for one gpurun call
mkdir -p gpurun_out
timeout 900 python - > gpurun_out/drift_L14.json 2> gpurun_out/drift_L14.err <<'PY'
import json, sys
sys.path.insert(0, ".")
from mcm_amd.parity import measure_drift, HEADLINE_PIXELS
print(json.dumps(measure_drift("ViT-L/14", n_id=20000, n_ood=5000, batch=256, arms=("fp16", "bf16"), **HEADLINE_PIXELS)))
PY
timeout 900 python - > gpurun_out/drift_B32.json 2> gpurun_out/drift_B32.err <<'PY'
import json, sys
sys.path.insert(0, ".")
from mcm_amd.parity import measure_drift, HEADLINE_PIXELS
print(json.dumps(measure_drift("ViT-B/32", n_id=50000, n_ood=10000, batch=512, arms=("fp16", "bf16"), **HEADLINE_PIXELS)))
PY
cat gpurun_out/drift_L14.json gpurun_out/drift_B32.json; tail -2 gpurun_out/drift_L14.err
