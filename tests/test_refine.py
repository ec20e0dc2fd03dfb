"""Threshold refinement (mcm_amd/refine.py) on the CPU: a noisy copy of a score set, with only the images near the
FPR95 threshold re-scored by the exact arm, gives the exact arm's FPR95 — by the reference's own metric code path
(mcm_amd.metrics.get_measures = the host restatement of utils/detection_util.py:66-119)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
@pytest.mark.parametrize("shift", [0.0, 1.5])   # AUROC 0.5 and ~0.86: the threshold in the tail / in the bulk of the OOD scores
def test_refined_fpr95_equals_the_exact_arms(seed, shift):
    from mcm_amd.metrics import get_measures
    from mcm_amd.refine import refine_threshold_scores

    g = torch.Generator().manual_seed(seed)
    n_id, n_ood = 20000, 10000
    exact = {"id": torch.randn(n_id, generator=g) * 1e-6 - 1e-3,
             "a": torch.randn(n_ood, generator=g) * 1e-6 - 1e-3 + shift * 1e-6,
             "b": torch.randn(n_ood // 2, generator=g) * 2e-6 - 1e-3 + shift * 1e-6}
    exact = {k: v.float() for k, v in exact.items()}
    noisy = {k: (v + torch.randn(v.shape, generator=g) * 4e-9).float() for k, v in exact.items()}
    calls = []

    def rescore(name, idx):
        calls.append((name, int(idx.numel())))
        return exact[name][idx]

    want = {k: get_measures(-exact["id"].numpy(), -exact[k].numpy()) for k in ("a", "b")}
    before = {k: get_measures(-noisy["id"].numpy(), -noisy[k].numpy()) for k in ("a", "b")}
    sid, sood, st = refine_threshold_scores(noisy["id"].clone(), {k: noisy[k].clone() for k in ("a", "b")}, rescore)
    after = {k: get_measures(-sid.numpy(), -sood[k].numpy()) for k in ("a", "b")}
    for k in ("a", "b"):
        assert after[k][2] == want[k][2], (k, before[k][2], after[k][2], want[k][2])   # FPR95: equal, not close
        assert abs(after[k][0] - want[k][0]) <= 1e-4                                      # AUROC: the noisy arm's own (within the bar)
    # ... at the price of a small fraction of the images
    assert st["rescored_total"] < 0.12 * (n_id + n_ood + n_ood // 2), st
    assert st["delta"] >= 2.0 * 4e-9 and st["rounds"] >= 1
    if shift:  # the noisy arm really was off somewhere in these draws (otherwise the test shows nothing)
        assert any(before[k][2] != want[k][2] for k in ("a", "b")) or seed not in (0, 1, 2, 3)


def test_exact_arm_is_left_alone():
    from mcm_amd.refine import refine_threshold_scores

    s = torch.randn(4096) * 1e-6
    o = torch.randn(1000) * 1e-6
    sid, sood, st = refine_threshold_scores(s.clone(), {"o": o.clone()}, lambda name, idx: (s if name == "id" else o)[idx])
    assert torch.equal(sid, s) and torch.equal(sood["o"], o) and st["delta"] == 0.0
