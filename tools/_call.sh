#!/bin/bash
# scratch driver (round 4, call 9): pixel-gathering patch GEMM with the pixel loads at the top of the step: tests + A/B
mkdir -p gpurun_out/r4c09
O=$PWD/gpurun_out/r4c09
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "patch_gemm or uint8" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="--no-drift --cpu-seconds 0 --no-arms --sustain-seconds 3 --ingest none"
for v in 1 0 1 0 1 0; do
timeout 300 python bench.py $B --patch-fold $v > $O/bench_pf$v.json 2> $O/bench_pf$v.err
python - <<PY
import json
d=json.loads(open("$O/bench_pf$v.json").read().strip().splitlines()[-1])
print("patch_fold $v", round(d["value"]), d["ms_per_step"], d["kernel_ms_per_step"])
PY
done
