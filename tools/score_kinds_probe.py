"""score_kinds_probe.py — AUROC / AUPR / FPR95 drift of the fp16 arm (raw and with threshold refinement) against the
exact-fp32 arm for every --score kind of the reference (utils/detection_util.py:233-248), B/16, K = 1000.

    python tools/score_kinds_probe.py [n_id n_ood]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mcm_amd.parity import measure_drift

    n_id, n_ood = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 10000)
    for score, T in (("MCM", 1.0), ("max-logit", 1.0), ("energy", 1.0), ("entropy", 1.0), ("var", 1.0), ("MCM", 0.01)):
        d = measure_drift("ViT-B/16", K=1000, n_id=n_id, n_ood=n_ood, batch=500, arms=("fp16", "fp16+refine"),
                          weights="fp16-exact", score=score, T=T)
        row = {"score": score, "T": T, "reference": {k: d["reference"][k] for k in ("auroc", "fpr95", "score_mean_id", "score_std_id")},
               "arms": {p: {k: v[k] for k in ("d_auroc", "d_aupr", "d_fpr95", "rms_dscore")} | {"images": v["max_set"]["d_fpr95_images"]}
                        for p, v in d["arms"].items()},
               "rescored": d["refine"]["fp16+refine"]["rescored_total"]}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
