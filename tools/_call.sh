#!/bin/bash
# scratch driver (round 3, call 48): LayerNorm tail on / off (harness library) at other checkpoints and batch sizes
mkdir -p gpurun_out/r3c48
O=$PWD/gpurun_out/r3c48
one() {
  tag=$1; shift
  timeout 600 python bench.py --no-drift --cpu-seconds 0 --sustain-seconds 0 "$@" > $O/b_$tag.json 2> $O/b_$tag.err || tail -3 $O/b_$tag.err
  python - <<PY
import json
d=json.load(open("$O/b_$tag.json"))
print("$tag", round(d["value"]), round(d["ms_per_step"],3), d["kernel_ms_per_step"]["gemm"], d["kernel_ms_per_step"]["layernorm"])
PY
}
for rep in 1 2; do for t in 0 1; do
  one L14_b256_tail${t}_$rep --ckpt ViT-L/14 --batch 256 --steps 20 --ln-tail $t
  one B32_b512_tail${t}_$rep --ckpt ViT-B/32 --batch 512 --steps 40 --ln-tail $t
  one B16_b128_tail${t}_$rep --batch 128 --steps 40 --ln-tail $t
  one B16_b768_tail${t}_$rep --batch 768 --steps 20 --ln-tail $t
done; done 2>&1 | tee $O/bench.txt
