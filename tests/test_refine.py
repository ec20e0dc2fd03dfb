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


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_two_level_refinement_reports_the_exact_arms_fpr95_image_for_image(seed):
    """Round 5: level 1 = a better-but-not-exact arm (the split-activation arm: noise ~1 ulp of the exact arm's score),
    level 2 = the exact arm on the handful of images within a few of THOSE noise widths of the threshold.  The result is the
    exact arm's FPR95 on every set, with orders of magnitude fewer exact re-scores than level 1 re-scores."""
    from mcm_amd.metrics import get_measures
    from mcm_amd.refine import refine_threshold_scores

    g = torch.Generator().manual_seed(100 + seed)
    n_id, n_ood = 50000, 10000
    exact = {"id": (torch.randn(n_id, generator=g) * 1.4e-6 - 1e-3).float(),
             "a": (torch.randn(n_ood, generator=g) * 1.4e-6 - 1e-3 + 1e-6).float(),
             "b": (torch.randn(n_ood, generator=g) * 1.4e-6 - 1e-3).float()}
    mid = {k: (v + torch.randn(v.shape, generator=g) * 3e-10).float() for k, v in exact.items()}     # two exact-grade arms
    noisy = {k: (v + torch.randn(v.shape, generator=g) * 5.6e-9).float() for k, v in exact.items()}  # the fp16 arm
    n1, n2 = [0], [0]

    def rescore(name, idx):
        n1[0] += int(idx.numel())
        return mid[name][idx]

    def rescore_exact(name, idx):
        n2[0] += int(idx.numel())
        return exact[name][idx]

    want = {k: get_measures(-exact["id"].numpy(), -exact[k].numpy()) for k in ("a", "b")}
    sid, sood, st = refine_threshold_scores(noisy["id"].clone(), {k: noisy[k].clone() for k in ("a", "b")}, rescore,
                                            rescore_exact=rescore_exact)
    for k in ("a", "b"):
        assert get_measures(-sid.numpy(), -sood[k].numpy())[2] == want[k][2], k
    assert st["delta2"] < 0.2 * st["delta"] and st["delta2"] > 0
    assert st["rescored_exact_total"] == n2[0] and st["rescored_exact_total"] <= 64 + 0.15 * st["rescored_total"], st
    assert st["rescored_total"] < 0.05 * (n_id + 2 * n_ood)


def test_rescorer_shards_the_window_by_the_loaders_index_ranges(monkeypatch):
    """world_size > 1: a rank re-scores only the window images of its own contiguous shard; the all-reduce of the zero-filled
    patches is the concatenation.  Simulated here without a process group: two 'ranks' run one after the other and their
    contributions are summed the way the all-reduce sums them."""
    from mcm_amd import dist as mdist
    from mcm_amd.refine import Rescorer

    n = 1000
    truth = torch.arange(n, dtype=torch.float32) * 0.5 + 3.0

    class Loader:
        dataset = list(range(n))

        def gather(self, idx):
            return torch.tensor(idx, dtype=torch.long)

    class Scorer:
        max_batch = 7

        def score_images(self, px, bank, T, score):
            return truth[px]

    idx = torch.tensor([3, 999, 500, 499, 0, 742, 250, 251])
    parts, counts = [], []
    for rank in range(4):
        monkeypatch.setattr(mdist, "world", lambda r=rank: (r, 4))
        monkeypatch.setattr(mdist, "all_reduce_sum", lambda t: t)   # collected below instead
        r = Rescorer(Scorer(), torch.zeros(1), {"id": Loader()}, 1.0, "MCM")
        parts.append(r("id", idx))
        counts.append(r.scored_here)
        lo, hi = mdist.shard_range(n, rank, 4)
        assert all((lo <= int(i) < hi) == (float(v) != 0.0) for i, v in zip(idx, parts[-1]))
    assert torch.equal(sum(parts), truth[idx])
    assert sum(counts) == idx.numel() and max(counts) <= 3   # nobody scores the whole window


def test_window_covers_the_order_statistic_the_reference_metric_actually_uses():
    """With few ID images the neighbouring ID order statistics lie many noise widths apart, and the reference's FPR@recall
    (utils/detection_util.py:66-105: argmin |recall - 0.95| over the reversed threshold arrays) counts the OOD scores below
    s_(k+1), not s_(k) (k = round(0.95 n)).  A window centred on s_(k) alone misses the OOD images whose noise carries them
    across s_(k+1): dense OOD scores, sparse ID scores, both levels."""
    from mcm_amd.metrics import get_measures
    from mcm_amd.refine import refine_threshold_scores

    bad_before = 0
    for seed in range(30):
        g = torch.Generator().manual_seed(1000 + seed)
        n_id, n_ood = 400, 40000
        exact = {"id": torch.randn(n_id, generator=g).float(), "o": (torch.randn(n_ood, generator=g) + 1.0).float()}
        mid = {k: (v + torch.randn(v.shape, generator=g) * 2e-6).float() for k, v in exact.items()}
        noisy = {k: (v + torch.randn(v.shape, generator=g) * 1e-4).float() for k, v in exact.items()}
        want = get_measures(-exact["id"].numpy(), -exact["o"].numpy())[2]
        bad_before += get_measures(-noisy["id"].numpy(), -noisy["o"].numpy())[2] != want
        for two in (False, True):
            sid, sood, st = refine_threshold_scores(noisy["id"].clone(), {"o": noisy["o"].clone()},
                                                    (lambda name, idx: mid[name][idx]) if two else (lambda name, idx: exact[name][idx]),
                                                    rescore_exact=(lambda name, idx: exact[name][idx]) if two else None, calib=64)
            assert get_measures(-sid.numpy(), -sood["o"].numpy())[2] == want, (seed, two, st)
            assert st["threshold_interval"][0] < st["threshold_interval"][1]
    assert bad_before >= 5   # the noisy arm really is off on a good share of these draws
