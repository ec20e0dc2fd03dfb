"""Drift of the 16-bit operand modes against the exact-fp32 arm on the headline configuration.

`north_star` asks for AUROC / FPR95 "matching the reference to 1e-4".  AUROC and FPR95 depend only
on the ORDER of the scores (reference utils/detection_util.py:66-119), so the statement has to be
measured where it is made: full-depth tower, K = 1000 prompts, ImageNet-sized ID set (50 000) against
a 10 000-image OOD set.  `measure_drift` scores the same device-generated ID / OOD images with every
arm (each arm encodes its own prompt bank, as a real run in that mode would), keeps all scores in
HBM, evaluates them with the device metric kernels (`mcm_measures`) and reports the differences to
the fp32 arm, which is itself pinned to the CPU oracle / HF at 7e-9 in score
(tests/test_gpu_model.py).  Used by bench.py (`--drift`) and tests/test_gpu_headline_parity.py.

`external=` closes the chain against the reference's own arithmetic on the SAME images: a dict of scorer
factories `name -> f(geo, state_dict, ids, mask, device) -> (pixels -> scores[b])`.  The callers that may
(tests, bench.py) pass the HF `CLIPModel` fp32 scorer of oracle/hf_reference.py running on the same device;
this module never imports it.  Every arm — the fp32 arm included — is then also reported against each
external scorer (`vs_external`).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

# The set bench.py and tests/test_gpu_headline_parity.py report: the same seeded pixels as the rest of the
# build (unit noise + class-conditional pattern), seeded weights rounded to fp16 for every arm — the
# situation of the reference's checkpoints, whose Linear / conv / projection weights were released in fp16
# and are therefore exact as fp16 MFMA operands.  See DESIGN.md §2 for the other regimes measured.
HEADLINE_PIXELS = dict(amp=1.5, tile=0.0, weights="fp16-exact")


def measure_drift(ckpt: str = "ViT-B/16", *, K: int = 1000, n_id: int = 50000, n_ood: int = 10000,
                  batch: int = 512, arms: Sequence[str] = ("bf16", "fp16"), ref: str = "fp32",
                  device: int = 0, score: str = "MCM", T: float = 1.0, amp: float = 1.5,
                  tile: float = 0.0, weights: str = "fp32", seed: int = 1,
                  external: Optional[Dict[str, Callable]] = None) -> Dict:
    """weights="fp16-exact": every parameter of the seeded state dict is rounded to the nearest fp16 value
    first (for ALL arms, the fp32 reference included) — the situation of the reference's checkpoints, whose
    Linear / conv / projection weights were trained and released in fp16, so an fp16 operand copy of them is
    lossless and only activation rounding separates the fp16 arm from the fp32 one."""
    import torch

    from .config import geometry
    from .engine import NativeCLIP
    from .synth import DevicePatternLoader, make_token_ids
    from .weights import synth_state_dict

    geo = geometry(ckpt)
    import numpy as np

    sd = synth_state_dict(geo, 0)
    if weights == "fp16-exact":
        sd = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    elif weights != "fp32":
        raise ValueError(weights)
    ids, mask = make_token_ids(K, seed=2)
    dev = torch.device("cuda", device)
    names = [ref] + [a for a in arms if a != ref]
    nets, banks = {}, {}
    ext = {}
    try:
        for name, factory in (external or {}).items():
            ext[name] = factory(geo, sd, ids, mask, dev)
        for p in names:
            nets[p] = NativeCLIP(geo, sd, device=device, precision=p, max_batch=batch,
                                 max_prompt_tokens=max(K * ids.shape[1], 77))
            banks[p] = nets[p].get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        scores = {p: {} for p in names + list(ext)}
        for tag, n, ood in (("id", n_id, False), ("ood", n_ood, True)):
            loader = DevicePatternLoader(n, geo.image_size, K, batch, dev, ood=ood, seed=seed, amp=amp, tile=tile)
            parts = {p: [] for p in names + list(ext)}
            for px, _ in loader:
                for p in names:
                    parts[p].append(nets[p].score_images(px, banks[p], T, score))
                for e, fn in ext.items():
                    parts[e].append(fn(px).to(device=dev, dtype=torch.float32).reshape(-1))
            for p in parts:
                scores[p][tag] = torch.cat(parts[p])
        out = {"ckpt": ckpt, "K": K, "n_id": n_id, "n_ood": n_ood, "batch": batch, "score": score,
               "T": T, "reference_arm": ref, "pixels": {"amp": amp, "tile": tile}, "weights": weights, "arms": {}}
        # fp16 activations that hit +-65504 anywhere in the run (sticky per-handle counters; 0 = none)
        out["fp16_saturation_events"] = {p: nets[p].saturation_count() for p in names if p == "fp16"}
        m_ref = nets[ref].measures(scores[ref]["id"], scores[ref]["ood"], negate=True)
        sid = scores[ref]["id"]
        out["reference"] = {"auroc": m_ref[0], "aupr": m_ref[1], "fpr95": m_ref[2],
                            "score_mean_id": float(sid.mean()), "score_std_id": float(sid.std()),
                            "score_mean_ood": float(scores[ref]["ood"].mean()),
                            "score_std_ood": float(scores[ref]["ood"].std())}
        def delta(p, q, m_p, m_q):
            d = torch.cat([(scores[p][t] - scores[q][t]).abs() for t in ("id", "ood")])
            return {"d_auroc": abs(m_p[0] - m_q[0]), "d_aupr": abs(m_p[1] - m_q[1]),
                    "d_fpr95": abs(m_p[2] - m_q[2]), "max_abs_dscore": float(d.max()),
                    "rms_dscore": float(d.pow(2).mean().sqrt())}

        meas = {ref: m_ref}
        for p in names[1:] + list(ext):
            meas[p] = nets[ref].measures(scores[p]["id"], scores[p]["ood"], negate=True)
        for p in names[1:]:
            m = meas[p]
            out["arms"][p] = {"auroc": m[0], "aupr": m[1], "fpr95": m[2], **delta(p, ref, m, m_ref)}
        if ext:
            out["external"] = {e: {"auroc": meas[e][0], "aupr": meas[e][1], "fpr95": meas[e][2]} for e in ext}
            out["reference"]["vs_external"] = {e: delta(ref, e, m_ref, meas[e]) for e in ext}
            for p in names[1:]:
                out["arms"][p]["vs_external"] = {e: delta(p, e, meas[p], meas[e]) for e in ext}
        return out
    finally:
        for n in nets.values():
            n.close()
        ext.clear()


if __name__ == "__main__":  # python -m mcm_amd.parity [n_id n_ood [amp]]
    import json
    import sys

    a = sys.argv[1:]
    kw = {}
    if len(a) >= 2:
        kw.update(n_id=int(a[0]), n_ood=int(a[1]))
    if len(a) >= 3:
        kw.update(amp=float(a[2]))
    if len(a) >= 4:
        kw.update(tile=float(a[3]))
    if len(a) >= 5:
        kw.update(weights=a[4])
    print(json.dumps(measure_drift(**kw)))
