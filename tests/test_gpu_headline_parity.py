"""North-star parity on the headline configuration: full CLIP-ViT-B/16, K = 1000 prompts, 50 000 ID + 10 000 OOD
images (ImageNet-1k vs one OOD set), the 16-bit operand modes against the exact-fp32 MFMA arm, which is itself
pinned to the CPU oracle and to HF (test_gpu_model.py).  AUROC / AUPR / FPR95 by the device metric kernels.
Numbers and the regimes they were measured in: DESIGN.md §2."""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_headline_drift_of_the_benchmarked_dtype():
    from bench import DEFAULT_PRECISION
    from mcm_amd.parity import HEADLINE_PIXELS, measure_drift

    d = measure_drift("ViT-B/16", K=1000, n_id=50000, n_ood=10000, batch=512, arms=("fp16", "bf16"),
                      **HEADLINE_PIXELS)
    print("headline drift:", json.dumps(d))
    ref = d["reference"]
    assert 0.02 < ref["auroc"] < 0.98 and 0.0 < ref["fpr95"] < 1.0          # non-degenerate operating point
    arm = d["arms"][DEFAULT_PRECISION]
    assert DEFAULT_PRECISION == "fp16"
    # the bar of BASELINE.json's north_star; FPR95's quantum at 10 000 OOD images is exactly 1e-4
    assert arm["d_auroc"] <= 1e-4, d
    assert arm["d_fpr95"] <= 1e-4 + 1e-12, d
    assert arm["d_aupr"] <= 1e-4, d
    # the text tower is fp32 in every mode, so bf16 differs only by its vision-side operand rounding; it is
    # the documented ~10x coarser arm and is bounded here so a regression is visible
    assert d["arms"]["bf16"]["d_auroc"] <= 3e-3, d
    assert arm["rms_dscore"] < d["arms"]["bf16"]["rms_dscore"]
