"""Distribution of the fp16 arm's AUROC / AUPR / FPR95 difference to the exact-fp32 arm over independent draws of the
headline sets (the FPR95 difference is a count of images crossing one threshold: 0, 1, 2 ... x 1e-4):
python tests/probes/drift_seeds.py [weights] [n_seeds] [tile]  ->  one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcm_amd.parity import measure_drift  # noqa: E402

weights = sys.argv[1] if len(sys.argv) > 1 else "fp16-exact"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tile = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
out = {"weights": weights, "tile": tile, "n_id": 50000, "n_ood": 10000, "draws": []}
for seed in range(1, n + 1):
    d = measure_drift("ViT-B/16", arms=("fp16",), amp=1.5, tile=tile, weights=weights, seed=seed)
    a = d["arms"]["fp16"]
    out["draws"].append({"seed": seed, "auroc_fp32": d["reference"]["auroc"], "d_auroc": a["d_auroc"], "d_aupr": a["d_aupr"],
                         "d_fpr95": a["d_fpr95"], "rms_dscore": a["rms_dscore"], "score_std_id": d["reference"]["score_std_id"]})
    print(out["draws"][-1], file=sys.stderr, flush=True)
print(json.dumps(out))
