#!/bin/bash
# scratch driver (round 3, call 30): XCD-aware deal of the attention workgroups (harness arm 10) vs shipped (1)
mkdir -p gpurun_out/r3c30
O=$PWD/gpurun_out/r3c30
timeout 600 python tools/attn_probe.py 50 > $O/probe.txt 2>&1; grep -E "L=197|L=257|L=50|bit-equal" $O/probe.txt | cut -c1-200
B="timeout 600 python bench.py --no-drift --cpu-seconds 0 --steps 40"
one() {
  $B --attn-variant $2 > $O/b_$1.json 2> $O/b_$1.err || tail -3 $O/b_$1.err
  python - <<PY
import json
d=json.load(open("$O/b_$1.json"))
print("$1", round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["kernel_ms_per_step"])
PY
}
for rep in 1 2; do one a1_$rep 1; one a10_$rep 10; done 2>&1 | tee $O/bench.txt
