#!/bin/bash
# LayerNorm fusion arms against the shipped step (23 LayerNorm launches of ~110 kernels), all through the harness library so that
# the control and the arms share one build:
#   ln_tail=1     LayerNorm computed by the residual GEMM's idle waves from L2 (round 3)
#   ln_fold=1     z = gamma o x + row moments in the producer, normalised in the consumer's epilogue (round 3)
#   ln_cluster=1  the full-row epilogue spread over a row panel's workgroups (round 6, R6.3): they exchange row moments, LayerNorm from registers
#   ln_cluster=1,ln_cluster_spin=4   the same, never waiting: segments whose partners are late go to a clean-up launch (R6.4)
#   ln_row=1      the literal full-row tile, 64 x N in one workgroup (R6.7); ln_row=2: three W stages
# Usage:  bash tools/ln_fusion_ab.sh [batches...] > profiles/r06_ln_fusion_ab.txt      (default: 512; round 5's sweep: 8 16 32 64 128 256)
# Prints: batch arm mode images_per_sec ms_per_step {layernorm, gemm, out-proj, fc2} ms per step; every arm twice, interleaved
# with the control, eager and (small batches) replayed from one hipGraph.
cd "$(dirname "$0")/.."
batches=${@:-512}
echo "batch arm mode images_per_sec ms_per_step ln_ms gemm_ms outproj_ms fc2_ms"
for b in $batches; do
  modes="eager"; [ "$b" -le 256 ] && modes="eager graph"
  for rep in 1 2; do
    for arm in ln_tail=0 ln_cluster=1 ln_cluster=1,ln_cluster_spin=4 ln_row=1 ln_row=3 ln_tail=1 ln_fold=1; do
      for mode in $modes; do
        g=""; [ "$mode" = graph ] && g="--graph"
        out=$(timeout 300 python bench.py --quick --batch $b --steps 40 --warmup 5 --harness $arm $g --detail /tmp/ln_ab_detail.json 2>/dev/null | tail -n 1)
        python - "$b" "$arm" "$mode" "$out" <<'PY'
import json, sys
b, arm, mode, out = sys.argv[1:5]
try:
    d = json.loads(out)
    k = d.get("kernel_ms_per_step") or {}
    print(b, "control" if arm == "ln_tail=0" else arm, mode, round(d["value"], 1), round(d["ms_per_step"], 4), k.get("layernorm"), k.get("gemm"),
          k.get("gemm_outproj"), k.get("gemm_fc2"))
except Exception as e:
    print(b, arm, mode, "failed", str(e)[:80], out[-200:])
PY
      done
    done
  done
done
