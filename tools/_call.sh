#!/bin/bash
# scratch driver (round 3, call 21): HF-on-device parity for the other two checkpoints / regimes
mkdir -p gpurun_out/r3c21
O=$PWD/gpurun_out/r3c21
for cfg in "ViT-B/32 50000 10000 512 fp16-exact B32_fp16exact" "ViT-B/32 50000 10000 512 fp32 B32_fp32w" "ViT-L/14 50000 10000 256 fp32 L14_fp32w"; do set -- $cfg
  timeout 1500 python tests/probes/hf_gpu_probe.py $1 $2 $3 $4 $5 fp16,bf16 > $O/$6.json 2> $O/$6.err
  python - <<PY
import json
d=json.load(open("$O/$6.json"))
f=lambda x:{k:float("%.3g"%v) for k,v in x.items()}
print("$6", round(d["seconds"]), "s | fp32 arm vs hf", f(d["reference"]["vs_external"]["hf"]))
for a in d["arms"]: print("    ", a, "vs hf", f(d["arms"][a]["vs_external"]["hf"]))
PY
done 2>&1 | tee $O/summary.txt
