"""North-star parity on the headline configuration: full CLIP-ViT-B/16, K = 1000 prompts, 50 000 ID + 10 000 OOD
images (ImageNet-1k vs one OOD set).  Every native arm is compared with (a) the exact-fp32 MFMA arm and (b) the
reference's own arithmetic on the SAME images: HF transformers `CLIPModel`, fp32 eager, running on this device
(oracle/hf_reference.py — the checker; BASELINE.json north_star: "identical AUROC/FPR95 to the HF-CLIP
reference").  AUROC / AUPR / FPR95 by the device metric kernels.  Both weight regimes are asserted: fp16-exact
seeded weights (the reference's checkpoints were released in fp16) and fp32-valued seeded weights.
Numbers and the regimes they were measured in: DESIGN.md §2."""
import json

import pytest

pytestmark = pytest.mark.gpu


def _external():
    from oracle.hf_reference import hf_available, hf_scorer_factory

    why = hf_available()
    if why is not None:
        pytest.skip(f"transformers unavailable: {why}")
    return {"hf": hf_scorer_factory()}


@pytest.mark.parametrize("weights", ["fp16-exact", "fp32"])
def test_headline_parity_vs_hf_reference(weights):
    """BASELINE config 3, the configuration the metric is quoted on: the ImageNet-1k-sized ID set once against the four
    OOD sets of the reference's default run, every set and the AVG row (the reference's CSV)."""
    from bench import DEFAULT_PRECISION
    from mcm_amd.parity import CONFIG3_OOD_SETS, HEADLINE_PIXELS, measure_drift

    assert DEFAULT_PRECISION == "fp16"
    # HF itself scores the 85 640 images in the fp16-exact regime (50 s of the test); in the fp32-valued regime the
    # suite compares against the exact-fp32 arm only — that arm equals HF to 4e-7 there too, measured by every default
    # bench.py run (parity.fp32_valued_weights.vs_hf) — to keep `pytest -m gpu` within a few minutes
    with_hf = weights == "fp16-exact"
    d = measure_drift("ViT-B/16", K=1000, n_id=50000, batch=512, arms=("fp16", "bf16"), ood_sets=CONFIG3_OOD_SETS,
                      amp=HEADLINE_PIXELS["amp"], tile=HEADLINE_PIXELS["tile"], weights=weights,
                      external=_external() if with_hf else None)
    print(f"headline parity ({weights} weights):", json.dumps(d))
    ref = d["reference"]
    assert 0.02 < ref["auroc"] < 0.98 and 0.0 < ref["fpr95"] < 1.0          # non-degenerate operating point
    if with_hf:  # (a) the exact-fp32 arm IS the HF computation, to the metric quantum, on every set
        r = ref["vs_external"]["hf"]
        assert r["d_auroc"] <= 1e-5 and r["d_aupr"] <= 1e-5 and r["d_fpr95"] <= 1e-4 + 1e-12, r
        assert r["rms_dscore"] <= 1e-9, r
        assert all(v["d_auroc"] <= 1e-5 and v["d_fpr95_images"] <= 1 for v in r["per_set"].values()), r
    # (b) the benchmarked dtype against HF: the bar of BASELINE.json's north_star on the AVG row.  AUROC and AUPR are
    # averages over 5e8 (ID, OOD) pairs per set and are held to 1e-4 on every set (measured <= 5e-5).  FPR95 of one set
    # is a COUNT — the OOD images on the ID side of one threshold, quantum 1e-4 at 10 000 images; on this stress set
    # (score spread 0.13 % of |score|) fp16's score noise moves 0 - 4 images across it depending on
    # the draw (profiles/r03_drift_seeds.json), so: the AVG row to 1e-4, every single set to at most 4 images.
    arm = d["arms"]["fp16"]
    for vs in (arm, arm["vs_external"]["hf"]) if with_hf else (arm,):
        assert vs["d_auroc"] <= 1e-4 and vs["d_aupr"] <= 1e-4, (weights, vs)
        assert vs["d_fpr95"] <= 1e-4 + 1e-12, (weights, vs)
        for name, v in vs["per_set"].items():
            assert v["d_auroc"] <= 1e-4 and v["d_aupr"] <= 1e-4 and v["d_fpr95_images"] <= 4, (weights, name, v)
    # bf16 (the dtype BASELINE configs 2/3/5 name) does NOT meet 1e-4 in either regime — measured 1.2e-4 /
    # 1.1e-3 in AUROC; bounded here so a regression is visible, and reported by bench.py
    b = d["arms"]["bf16"]["vs_external"]["hf"] if with_hf else d["arms"]["bf16"]
    assert b["d_auroc"] <= 3e-3 and b["d_fpr95"] <= 3e-3, b
    assert arm["rms_dscore"] < d["arms"]["bf16"]["rms_dscore"]
    assert d["fp16_saturation_events"] == {"fp16": 0}, d["fp16_saturation_events"]  # nothing left the fp16 range


def test_l14_parity_vs_hf_reference():
    """BASELINE config 4 (ViT-L/14 fp16, batch 256) with 10 000 OOD images, so that FPR95's quantum is 1e-4
    (round 2 ran 5 000: one sample = 2e-4).  The full 50 000 + 10 000 run is profiles/r03_parity_L14_50k_vs_hf.json
    (fp16 vs HF: dAUROC 1.2e-5, dFPR95 0; 4 minutes); here 4 000 + 10 000 (50 s).  FPR95 as an image count, see
    above: measured 0 and 4 images on two draws of a 20 000 + 10 000 set."""
    from mcm_amd.parity import HEADLINE_PIXELS, measure_drift

    d = measure_drift("ViT-L/14", K=1000, n_id=4000, n_ood=10000, batch=256, arms=("fp16",),
                      amp=HEADLINE_PIXELS["amp"], tile=HEADLINE_PIXELS["tile"], weights="fp16-exact",
                      external=_external())
    print("L/14 parity (fp16-exact weights):", json.dumps(d))
    r = d["reference"]["vs_external"]["hf"]
    assert r["d_auroc"] <= 1e-5 and r["d_fpr95"] <= 1e-4 + 1e-12, r
    for vs in (d["arms"]["fp16"], d["arms"]["fp16"]["vs_external"]["hf"]):
        assert vs["d_auroc"] <= 1e-4 and vs["d_aupr"] <= 1e-4 and vs["d_fpr95"] <= 5e-4 + 1e-12, vs


@pytest.mark.parametrize("weights", ["fp16-exact", "fp32"])
def test_config2_parity_vs_hf_k100(weights):
    """BASELINE config 2's sizes (ImageNet-100 ID vs one 10 000-image OOD set: K = 100, 5 000 + 10 000 images) against
    HF on this device, both weight regimes.  Measured (profiles/r03_k100_parity.txt): fp16 dAUROC 3.4e-5 / 5.1e-5,
    dFPR95 1e-4 / 0; bf16 — the dtype the config names — 5.7e-4 / 5.3e-4: it does not meet the bar."""
    from mcm_amd.parity import HEADLINE_PIXELS, measure_drift

    d = measure_drift("ViT-B/16", K=100, n_id=5000, n_ood=10000, batch=512, arms=("fp16", "bf16"),
                      amp=HEADLINE_PIXELS["amp"], tile=HEADLINE_PIXELS["tile"], weights=weights, external=_external())
    print(f"config-2-sized parity ({weights} weights):", json.dumps(d))
    r = d["reference"]["vs_external"]["hf"]
    assert r["d_auroc"] <= 1e-5 and r["d_fpr95"] <= 1e-4 + 1e-12 and r["rms_dscore"] <= 5e-9, r
    vs = d["arms"]["fp16"]["vs_external"]["hf"]
    assert vs["d_auroc"] <= 1e-4 and vs["d_aupr"] <= 1e-4 and vs["d_fpr95"] <= 4e-4 + 1e-12, vs
    b = d["arms"]["bf16"]["vs_external"]["hf"]
    assert b["d_auroc"] <= 3e-3 and b["d_fpr95"] <= 3e-3, b
