"""outlier_probe.py — the outlier-channel stress checkpoint (mcm_amd/weights.py::inject_outlier_channels) through every arm:
score drift, AUROC / FPR95 differences against the exact-fp32 arm, fp16 saturation events.

    python tools/outlier_probe.py [n_id n_ood]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mcm_amd.config import geometry
    from mcm_amd.parity import measure_drift
    from mcm_amd.weights import inject_outlier_channels, synth_state_dict

    n_id, n_ood = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 5000)
    geo = geometry("ViT-B/16")
    base = synth_state_dict(geo, 0, "fp16-exact")
    for scale, gs in ((30.0, 1.0), (100.0, 1.0), (200.0, 1.0), (30.0, None), (100.0, None), (200.0, None)):
        sd, ch = inject_outlier_channels(base, geo, channels=6, scale=scale, gamma_scale=gs)
        d = measure_drift("ViT-B/16", K=1000, n_id=n_id, n_ood=n_ood, batch=500, arms=("fp16", "bf16"), state_dict=sd)
        row = {"scale": scale, "gamma_scale": scale if gs is None else gs, "channels": ch.tolist(),
               "reference": d["reference"], "saturation": d["fp16_saturation_events"], "weight_operands": d["weight_operands"],
               "arms": {p: {k: v[k] for k in ("d_auroc", "d_aupr", "d_fpr95", "rms_dscore", "max_abs_dscore")}
                        for p, v in d["arms"].items()}}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
