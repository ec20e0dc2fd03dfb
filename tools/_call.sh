#!/bin/bash
# scratch driver (round 4, call 12): parity of every --score kind; batch sweep and other checkpoints on the shipped tree
mkdir -p gpurun_out/r4c12
O=$PWD/gpurun_out/r4c12
timeout 900 python tools/score_kinds_probe.py 10000 10000 > $O/score_kinds.txt 2> $O/score_kinds.err; cat $O/score_kinds.txt | cut -c1-600
B="--no-drift --cpu-seconds 0 --no-arms --sustain-seconds 0 --ingest none"
for b in 8 16 32 64 128 256 768; do
timeout 300 python bench.py $B --batch $b > $O/bench_b$b.json 2> $O/bench_b$b.err
python - <<PY
import json
d=json.loads(open("$O/bench_b$b.json").read().strip().splitlines()[-1])
print("batch $b", round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), d["kernel_ms_per_step"])
PY
done
for c in "ViT-B/32 512 fp16" "ViT-L/14 256 fp16" "ViT-B/16 512 bf16"; do set -- $c
timeout 300 python bench.py $B --ckpt $1 --batch $2 --precision $3 --weight-operands single > $O/bench_$(echo $1 | tr / _)_$3.json 2> $O/bench_other.err
python - <<PY
import json
d=json.loads(open("$O/bench_$(echo $1 | tr / _)_$3.json").read().strip().splitlines()[-1])
print("$1 batch $2 $3", round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), d["kernel_ms_per_step"])
PY
done
