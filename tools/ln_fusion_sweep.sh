#!/bin/bash
# LayerNorm fusion arms at small batches: the shipped step (23 LayerNorm launches of ~110 kernels), the LayerNorm
# tail (computed by the residual GEMM's idle waves: needs the ping-pong kernel, i.e. batches the persistent kernel takes) and the
# LayerNorm fold (z = gamma o x + row moments, normalised in the consumer's epilogue), eager and replayed from one hipGraph.
# Prints: batch arm mode img/s ms/step layernorm-launches-per-step.   Usage: bash tools/ln_fusion_sweep.sh > profiles/r05_ln_fusion_sweep.txt
cd "$(dirname "$0")/.."
echo "batch arm mode images_per_sec ms_per_step ln_ms_per_step gemm_ms_per_step"
for b in 8 16 32 64 128 256; do
  for arm in base ln_tail=1 ln_fold=1; do
    for mode in eager graph; do
      h=""; [ "$arm" != base ] && h="--harness $arm"
      g=""; [ "$mode" = graph ] && g="--graph"
      out=$(python bench.py --quick --batch $b --steps 40 --warmup 5 $h $g 2>/dev/null | tail -1)
      python - "$b" "$arm" "$mode" "$out" <<'PY'
import json, sys
b, arm, mode, out = sys.argv[1:5]
try:
    d = json.loads(out)
    k = d.get("kernel_ms_per_step") or {}
    print(b, arm, mode, round(d["value"], 1), round(d["ms_per_step"], 4), k.get("layernorm"), k.get("gemm"))
except Exception as e:
    print(b, arm, mode, "failed", str(e)[:80], out[-200:])
PY
    done
  done
done
