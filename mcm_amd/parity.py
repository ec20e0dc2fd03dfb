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

# BASELINE config 3 — the configuration the metric is quoted on: ImageNet-1k ID (50 000 images) against the four OOD
# sets of the reference's default run, (name, size, seed): eval_ood_detection.py:63-68, sizes SURVEY.md §8a-A9.
# `measure_drift(ood_sets=CONFIG3_OOD_SETS)` scores the ID set once and reports every OOD set and the AVG row, like
# the reference's CSV.
CONFIG3_OOD_SETS = (("iNaturalist", 10000, 11), ("SUN", 10000, 12), ("places365", 10000, 13), ("dtd", 5640, 14))


# A second operating point.  What a random-init tower can be made to give (tools/spread_probe.py — removed in round 6, git history —,
# profiles/r04_b_spread_probe.txt): a strong per-class texture (`tile`) widens the per-image score spread from 0.13 % to
# 0.6 % of |score| (std 1.4e-6 -> 6.5e-6; it saturates there — at T = 1 and K = 1000 the softmax is nearly flat,
# d score / d cos = 1/K, so even a real checkpoint's cosines spread the score by only a few 1e-5), which puts fp16's
# score noise at ~0.1 % of the spread instead of 0.4 %: a real checkpoint's ratio.  What no pixel regime gives a random
# tower is SEPARATION (AUROC stays 0.5 +- 0.02): `operating_point=0.9` therefore builds the ID / OOD split from the
# reference arm's own scores — image i of the scored pool is called ID with probability sigmoid(-a z_i), z the
# standardised reference score, `a` found by bisection so that the reference's AUROC is the target — and every arm is
# then evaluated on that same split of the same images.
REALISTIC_PIXELS = dict(amp=1.5, tile=30.0, tile_ood=30.0, weights="fp16-exact")


def _arm_spec(arm: str):
    """"fp16" | "bf16" | "fp32", optionally ":single" / ":split" (the weight-operand form, include/mcm.h MCM_WEIGHTS_*;
    default "auto": split exactly when a GEMM weight is not a number of the operand dtype)."""
    prec, _, wo = arm.partition(":")
    return prec, (wo or "auto")


def measure_drift(ckpt: str = "ViT-B/16", *, K: int = 1000, n_id: int = 50000, n_ood: int = 10000,
                  batch: int = 512, arms: Sequence[str] = ("bf16", "fp16"), ref: str = "fp32",
                  device: int = 0, score: str = "MCM", T: float = 1.0, amp: float = 1.5,
                  tile: float = 0.0, weights: str = "fp32", seed: int = 1,
                  external: Optional[Dict[str, Callable]] = None,
                  ood_sets: Optional[Sequence] = None, tile_ood: Optional[float] = None,
                  state_dict: Optional[Dict] = None, operating_point: Optional[float] = None,
                  feature_cache: Optional[Dict] = None) -> Dict:
    """weights="fp16-exact": every parameter of the seeded state dict is rounded to the nearest fp16 value
    first (for ALL arms, the fp32 reference included) — the situation of the reference's checkpoints, whose
    Linear / conv / projection weights were trained and released in fp16, so an fp16 operand copy of them is
    lossless and only activation rounding separates the fp16 arm from the fp32 one.  weights="fp32": as drawn; a 16-bit
    arm then runs the split-weight GEMMs (arm "fp16") unless it says "fp16:single".
    ood_sets = ((name, size, seed), ...): several OOD sets against the one ID set; d_auroc / d_aupr / d_fpr95 of an
    arm are then the differences of the AVG row (mean of the per-set metrics), `per_set` holds each set's own and
    `max_set` the largest per-set difference (what "identical FPR95 per dataset" is judged on).
    tile_ood: texture strength of the OOD sets when it differs from the ID set's `tile`.
    state_dict: score with these parameters instead of the seeded ones (`weights` then only labels the result).
    operating_point: also report every arm on an ID / OOD split of ALL scored images that gives the reference arm this
    AUROC (`out["operating_point"]`; see REALISTIC_PIXELS).
    feature_cache: a dict the caller keeps between calls that differ ONLY in `score` / `T`: the unit-norm image features of
    every arm and set are computed by the first call and re-used by the others (mcm_score IS mcm_encode_image followed by
    mcm_score_features on the same buffer: same bits), so five score kinds cost one pass through the towers."""
    import torch

    from .config import geometry
    from .engine import NativeCLIP
    from .synth import DevicePatternLoader, make_token_ids
    from .weights import synth_state_dict

    geo = geometry(ckpt)
    if state_dict is not None:
        sd = state_dict
    elif weights in ("fp16-exact", "fp32"):
        sd = synth_state_dict(geo, 0, weights)
    else:
        raise ValueError(weights)
    ids, mask = make_token_ids(K, seed=2)
    dev = torch.device("cuda", device)
    # derived arms (mcm_amd/refine.py): "fp16+refine" = arm "fp16" with the images near the FPR95 threshold re-scored by the
    # split-activation arm "fp16x2" (what the CLI does by default); "fp16+refine2" = additionally the inner window by the
    # reference arm (the CLI's --refine-threshold exact).  A 16-bit arm without a split-activation form (bf16) is re-scored
    # by the reference arm directly.
    derived = [a for a in arms if a.endswith("+refine") or a.endswith("+refine2")]
    base_arms = [a for a in arms if a not in derived]
    for a in derived:
        b = a.partition("+")[0]
        if b not in base_arms:
            base_arms.append(b)
        if _arm_spec(b)[0] == "fp16" and b + "x2" not in base_arms:
            base_arms.append(b + "x2")
    names = [ref] + [a for a in base_arms if a != ref]
    nets, banks, scorers = {}, {}, {}
    ext = {}
    try:
        for name, factory in (external or {}).items():
            ext[name] = factory(geo, sd, ids, mask, dev)
        for p in names:
            if p.endswith("x2"):  # "fp16x2": the split-activation arm of the "fp16" handle (same weights, same workspace)
                continue
            prec, wo = _arm_spec(p)
            nets[p] = NativeCLIP(geo, sd, device=device, precision=prec, max_batch=batch, weight_operands=wo,
                                 max_prompt_tokens=max(K * ids.shape[1], 77))
            banks[p] = nets[p].get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
            scorers[p] = nets[p].score_images
        for p in names:
            if p.endswith("x2"):
                b = p[:-2]
                if b not in nets:
                    prec, wo = _arm_spec(b)
                    nets[b] = NativeCLIP(geo, sd, device=device, precision=prec, max_batch=batch, weight_operands=wo,
                                         max_prompt_tokens=max(K * ids.shape[1], 77))
                    banks[b] = nets[b].get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
                banks[p], scorers[p] = banks[b], nets[b].score_images_x2
        # one OOD set (the default) or several: BASELINE config 3 scores the ID set once against four OOD sets and
        # reports every set plus their average (the reference's CSV, eval_ood_detection.py:86-98)
        sets = [("ood", n_ood, seed)] if not ood_sets else [(str(n), int(c), int(sd_)) for n, c, sd_ in ood_sets]
        tags = ["id"] + [n for n, _, _ in sets]
        scores = {p: {} for p in names + list(ext)}
        t_ood = tile if tile_ood is None else tile_ood
        for tag, n, ood, sd_ in [("id", n_id, False, seed)] + [(n, c, True, s_) for n, c, s_ in sets]:
            loader = DevicePatternLoader(n, geo.image_size, K, batch, dev, ood=ood, seed=sd_, amp=amp,
                                         tile=t_ood if ood else tile)
            parts = {p: [] for p in names + list(ext)}
            if feature_cache is not None and not ext:
                if tag not in feature_cache:
                    feats = {p: [] for p in names}
                    for px, _ in loader:
                        for p in names:
                            b = p[:-2] if p.endswith("x2") else p
                            fn = nets[b].get_image_features_x2 if p.endswith("x2") else nets[b].get_image_features
                            feats[p].append(fn(px, normalize=True))
                    feature_cache[tag] = {p: torch.cat(v) for p, v in feats.items()}
                for p in names:
                    b = p[:-2] if p.endswith("x2") else p
                    parts[p].append(nets[b].score_features(feature_cache[tag][p], banks[p], T, score))
            else:
                for px, _ in loader:
                    for p in names:
                        parts[p].append(scorers[p](px, banks[p], T, score))
                    for e, fn in ext.items():
                        parts[e].append(fn(px).to(device=dev, dtype=torch.float32).reshape(-1))
            for p in parts:
                scores[p][tag] = torch.cat(parts[p])
        # derived arms: the base arm's scores with the images near the FPR95 threshold re-scored by the exact arm (here the
        # exact arm has scored everything already, so the callback is a lookup)
        refine_stats = {}
        if derived:
            from .refine import refine_threshold_scores

            for a in derived:
                b, _, kind = a.partition("+")
                first = b + "x2" if b + "x2" in scores else ref
                scores[a] = {t: scores[b][t].clone() for t in tags}
                _, _, st = refine_threshold_scores(scores[a]["id"], {n: scores[a][n] for n, _, _ in sets},
                                                   lambda name, idx, first=first: scores[first][name][idx],
                                                   rescore_exact=(lambda name, idx: scores[ref][name][idx])
                                                   if (kind == "refine2" and first != ref) else None)
                # what the refinement's argument assumes of the images it did NOT calibrate on, measured on every image of every
                # set: the largest |re-scorer - base arm| score difference (mcm_amd/refine.py: delta = margin x the calibration max)
                noise_all = max(float((scores[first][t] - scores[b][t]).abs().max()) for t in tags)
                refine_stats[a] = dict(st, rescorer=first, noise_all_max_abs=noise_all, calibration_bound_held=bool(noise_all <= st["delta"]))
            names = names + derived
        out = {"ckpt": ckpt, "K": K, "n_id": n_id, "n_ood": n_ood if not ood_sets else {n: c for n, c, _ in sets},
               "batch": batch, "score": score, "T": T, "reference_arm": ref,
               "pixels": {"amp": amp, "tile": tile, "tile_ood": t_ood},
               "weights": weights if state_dict is None else "caller's state dict", "arms": {}}
        # fp16 activations that hit +-65504 anywhere in the run (sticky per-handle counters; 0 = none)
        out["fp16_saturation_events"] = {p: nets[p].saturation_count() for p in nets if _arm_spec(p)[0] == "fp16"}
        out["weight_operands"] = {p: {"split": nets[p].split_weights, "inexact_elements": nets[p].weights_inexact}
                                  for p in nets if p in names}
        if refine_stats:
            out["refine"] = refine_stats

        def measures(p):  # per OOD set, and the AVG row (mean over the sets) — what the metrics are quoted on
            per = {n: nets[ref].measures(scores[p]["id"], scores[p][n], negate=True) for n, _, _ in sets}
            avg = tuple(sum(m[i] for m in per.values()) / len(per) for i in range(3))
            return per, avg

        def delta(p, q):
            (pp, pa), (qp, qa) = meas[p], meas[q]
            d = torch.cat([(scores[p][t] - scores[q][t]).abs() for t in tags])
            r = {"d_auroc": abs(pa[0] - qa[0]), "d_aupr": abs(pa[1] - qa[1]), "d_fpr95": abs(pa[2] - qa[2]),
                 "max_abs_dscore": float(d.max()), "rms_dscore": float(d.pow(2).mean().sqrt())}
            per = {n: {"d_auroc": abs(pp[n][0] - qp[n][0]), "d_aupr": abs(pp[n][1] - qp[n][1]),
                       "d_fpr95": abs(pp[n][2] - qp[n][2]),
                       "d_fpr95_images": round(abs(pp[n][2] - qp[n][2]) * c)} for n, c, _ in sets}
            # the largest per-set difference: opposite-sign drifts of different sets cancel in the AVG row, not here
            r["max_set"] = {k: max(v[k] for v in per.values()) for k in ("d_auroc", "d_aupr", "d_fpr95", "d_fpr95_images")}
            if len(sets) > 1:  # the top-level keys are the AVG row; every set on its own, FPR95 also as an image count
                r["per_set"] = per
            return r

        meas = {p: measures(p) for p in names + list(ext)}
        m_ref = meas[ref][1]
        sid = scores[ref]["id"]
        sood = torch.cat([scores[ref][n] for n, _, _ in sets])
        out["reference"] = {"auroc": m_ref[0], "aupr": m_ref[1], "fpr95": m_ref[2],
                            "score_mean_id": float(sid.mean()), "score_std_id": float(sid.std()),
                            "score_mean_ood": float(sood.mean()), "score_std_ood": float(sood.std())}
        if len(sets) > 1:
            out["reference"]["per_set"] = {n: dict(zip(("auroc", "aupr", "fpr95"), meas[ref][0][n])) for n, _, _ in sets}
        for p in names[1:]:
            m = meas[p][1]
            out["arms"][p] = {"auroc": m[0], "aupr": m[1], "fpr95": m[2], **delta(p, ref)}
        if ext:
            out["external"] = {e: dict(zip(("auroc", "aupr", "fpr95"), meas[e][1])) for e in ext}
            out["reference"]["vs_external"] = {e: delta(ref, e) for e in ext}
            for p in names[1:]:
                out["arms"][p]["vs_external"] = {e: delta(p, e) for e in ext}
        if operating_point is not None:
            out["operating_point"] = _operating_point(nets[ref], {p: torch.cat([scores[p][t] for t in tags])
                                                                  for p in names + list(ext)}, ref, float(operating_point),
                                                      list(ext))
        return out
    finally:
        for n in nets.values():
            n.close()
        ext.clear()


def _operating_point(net, pool: Dict, ref: str, target: float, ext: Sequence[str], seed: int = 97) -> Dict:
    """ID / OOD split of a scored pool that gives the reference arm AUROC = target (see REALISTIC_PIXELS), and every
    arm's metrics on that split.  `pool[p]` = scores of arm p for the same images, in the same order."""
    import torch

    s_ref = pool[ref].double()
    z = ((s_ref - s_ref.mean()) / s_ref.std()).float()
    u = torch.rand(z.numel(), generator=torch.Generator(device=z.device).manual_seed(seed), device=z.device)

    def split(a):  # scores are negated confidences: the lower, the more ID
        return u < torch.sigmoid(-a * z)

    def auroc_of(a):
        m = split(a)
        return net.measures(pool[ref][m], pool[ref][~m], negate=True)[0]

    lo, hi = 0.0, 64.0
    for _ in range(40):  # AUROC(a) rises from 0.5 monotonically
        mid = 0.5 * (lo + hi)
        if auroc_of(mid) < target:
            lo = mid
        else:
            hi = mid
    a = 0.5 * (lo + hi)
    m = split(a)
    n_id, n_ood = int(m.sum()), int((~m).sum())
    refine_stats = {}
    for p in [p for p in pool if p.endswith("+refine") or p.endswith("+refine2")]:  # threshold refinement on THIS split
        from .refine import refine_threshold_scores

        b, _, kind = p.partition("+")
        first = b + "x2" if b + "x2" in pool else ref
        sid, sood = pool[b][m].clone(), pool[b][~m].clone()
        fid, food = pool[first][m], pool[first][~m]
        rid, rood = pool[ref][m], pool[ref][~m]
        _, _, st = refine_threshold_scores(sid, {"ood": sood}, lambda name, idx: (fid if name == "id" else food)[idx],
                                           rescore_exact=(lambda name, idx: (rid if name == "id" else rood)[idx])
                                           if (kind == "refine2" and first != ref) else None)
        refine_stats[p] = dict(st, rescorer=first)
        patched = pool[b].clone()
        patched[m], patched[~m] = sid, sood
        pool[p] = patched
    meas = {p: net.measures(pool[p][m], pool[p][~m], negate=True) for p in pool}
    r = meas[ref]
    out = {"target_auroc": target, "a": a, "n_id": n_id, "n_ood": n_ood,
           "reference": {"auroc": r[0], "aupr": r[1], "fpr95": r[2], "score_std": float(pool[ref].std()),
                         "score_mean": float(pool[ref].mean())}, "arms": {}}

    def delta(p, q):
        return {"d_auroc": abs(meas[p][0] - meas[q][0]), "d_aupr": abs(meas[p][1] - meas[q][1]),
                "d_fpr95": abs(meas[p][2] - meas[q][2]), "d_fpr95_images": round(abs(meas[p][2] - meas[q][2]) * n_ood),
                "rms_dscore": float((pool[p] - pool[q]).double().pow(2).mean().sqrt())}

    for p in pool:
        if p == ref or p in ext:
            continue
        out["arms"][p] = {"auroc": meas[p][0], "fpr95": meas[p][2], **delta(p, ref)}
        if ext:
            out["arms"][p]["vs_external"] = {e: delta(p, e) for e in ext}
    if ext:
        out["reference"]["vs_external"] = {e: delta(ref, e) for e in ext}
    if refine_stats:
        out["refine"] = refine_stats
    return out


def meets_bar(d: Dict, bar: float = 1e-4, fpr_images: int = 1) -> bool:
    """north_star's tolerance on one `delta` record, judged per OOD set (not on the AVG row, where opposite-sign drifts
    cancel): |dAUROC|, |dAUPR| <= bar on every set, FPR95 within `fpr_images` images of the reference on every set (its
    quantum: one image of a 10 000-image set IS 1e-4)."""
    m = d["max_set"]
    return bool(m["d_auroc"] <= bar and m["d_aupr"] <= bar and m["d_fpr95_images"] <= fpr_images)


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
