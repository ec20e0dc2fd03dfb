"""x2_parity_probe.py — the split-activation arm (mcm_score_x2) and the two refinement forms against the exact-fp32 arm at full
set sizes on the other geometries and regimes (the headline configuration is bench.py's `parity` leg): ViT-B/32 and ViT-L/14,
fp16-exact and fp32-valued weights, K = 100 at config 2's sizes, and the outlier-channel checkpoint.
Usage (GPU box):  python tools/x2_parity_probe.py > profiles/r05_f_x2_parity_other_checkpoints.txt"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.parity import CONFIG3_OOD_SETS, HEADLINE_PIXELS, measure_drift  # noqa: E402
from mcm_amd.weights import inject_outlier_channels, synth_state_dict  # noqa: E402

ARMS = ("fp16", "fp16x2", "fp16+refine", "fp16+refine2")
KEYS = ("max_abs_dscore", "rms_dscore")


def show(tag, d):
    row = {a: {**{k: float(f"{v[k]:.3g}") for k in KEYS}, "d_auroc_max_set": float(f"{v['max_set']['d_auroc']:.3g}"),
               "fpr95_images_max_set": v["max_set"]["d_fpr95_images"]} for a, v in d["arms"].items()}
    st = d.get("refine", {})
    print(tag, json.dumps(row), json.dumps({a: {k: s.get(k) for k in ("rescored_total", "rescored_exact_total", "delta", "delta2")}
                                            for a, s in st.items()}), flush=True)


def main():
    px = dict(amp=HEADLINE_PIXELS["amp"], tile=HEADLINE_PIXELS["tile"])
    for ckpt, batch, n_id, sets in (("ViT-B/32", 512, 50000, CONFIG3_OOD_SETS), ("ViT-L/14", 256, 6000, (("ood", 10000, 11),))):
        for w in ("fp16-exact", "fp32"):
            show(f"{ckpt} {w} K=1000 n_id={n_id}:", measure_drift(ckpt, K=1000, n_id=n_id, batch=batch, arms=ARMS, ood_sets=sets,
                                                                    weights=w, **px))
    for w in ("fp16-exact", "fp32"):
        show(f"ViT-B/16 {w} K=100 (config 2 sizes):", measure_drift("ViT-B/16", K=100, n_id=5000, batch=512, arms=ARMS,
                                                                    ood_sets=CONFIG3_OOD_SETS, weights=w, **px))
    geo = geometry("ViT-B/16")
    sd, _ = inject_outlier_channels(synth_state_dict(geo, 0, "fp16-exact"), geo, channels=6, scale=100.0, gamma_scale=1.0)
    show("ViT-B/16 outlier-channel checkpoint:", measure_drift("ViT-B/16", K=1000, n_id=20000, n_ood=10000, batch=500, arms=ARMS,
                                                               state_dict=sd))


if __name__ == "__main__":
    main()
