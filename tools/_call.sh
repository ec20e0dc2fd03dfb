#!/bin/bash
# scratch driver (round 3, call 60): 64x128 tile kernel (harness variant 11) — bitwise test, then batches 8 ... 48 against the shipped choice
mkdir -p gpurun_out/r3c60
O=$PWD/gpurun_out/r3c60
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k tile64 > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for b in 8 16; do for v in -1 11; do
  timeout 100 python bench.py --batch $b --no-drift --cpu-seconds 0 --sustain-seconds 0 --steps 60 --attn-variant 1 --gemm-variant $v > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print("batch $b variant $v", round(d["value"]), round(d["ms_per_step"],3), d["kernel_ms_per_step"]["gemm"])
PY
done; done 2>&1 | tee $O/sweep.txt
