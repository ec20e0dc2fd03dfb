#!/bin/bash
# scratch driver (round 3, call 54): sliver split up to a quarter (shipped) vs half a round left over, batches 64 ... 512
mkdir -p gpurun_out/r3c54
O=$PWD/gpurun_out/r3c54
for b in 64 128 192 256 320 384 448 512; do for lib in libmcm_hip.so libmcm_hip_s2.so; do
  timeout 600 python tools/bench_with_lib.py mcm_amd/$lib --batch $b --no-drift --cpu-seconds 0 --sustain-seconds 0 --steps 40 > $O/b.json 2> $O/b.err || tail -3 $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print("batch $b $lib", round(d["value"]), round(d["ms_per_step"],3), d["kernel_ms_per_step"]["gemm"])
PY
done; done 2>&1 | tee $O/sweep.txt
