"""Threshold refinement: FPR@recall of a 16-bit arm made equal to the exact arm's, at the cost of re-scoring a few hundred images.

FPR95 (reference utils/detection_util.py:66-105) is a count: the OOD images on the ID side of ONE threshold, the score
below which 95 % of the ID set lies.  A 16-bit arm's score noise (rms 5.6e-9 for fp16 at B/16) is far smaller than the
scores' spread, so it can only change the count through the images whose score is within a few noise widths of that
threshold: 0 - 2 of 10 000 on the headline sets, 8 of 31 000 at a realistic operating point —
small, but not the reference's number.  Those images can be named: everything within `delta` of the threshold.  This
module re-scores exactly them with a better arm and patches their scores in place; every other image is on the same side
of the threshold in both arms AS LONG AS its own noise is below `delta` — which is an assumption about the 85 000 images
that were not calibrated, not a proof: `delta` is `margin` x the largest difference seen on `calib` images, a tail-probability
argument.  What holds it up is measured, not derived: tests/test_gpu_headline_parity.py asserts, over every image of BASELINE
config 3, max |split-activation arm - 16-bit arm| <= delta (the calibration bound held everywhere).  And the repair is
specific to FPR at THIS recall level: a score outside the window keeps its 16-bit noise, so another consumer of the
per-sample scores (another recall level, a downstream threshold) gets the raw arm there.

    delta = margin x (largest |better arm - 16-bit| score difference over a calibration sample of the ID set)

  1. calibration: `calib` ID images (runs of 64 spread evenly over the set) are re-scored -> noise estimate, delta;
  2. ID window: the ID images within delta of the provisional threshold are re-scored and the threshold recomputed
     (repeated if it moved by more than delta / 2);
  3. OOD windows: in every OOD set the images within delta of the final threshold are re-scored.

Round 5: the re-scorer (`rescore`) is the SPLIT-ACTIVATION arm of the scoring handle itself (include/mcm.h mcm_score_x2:
hi + lo fp16 operands, within one fp32 ulp of the exact-fp32 arm's score, ~5 x its throughput, no second model in HBM).  Two
exact-grade arms still differ by their own round-off (HF fp32 vs the fp32 MFMA arm: 1 image of 10 000 now and then), so
an optional second level (`rescore_exact`, the exact-fp32 arm on a small handle) re-scores the handful of images within
delta2 = margin x max |exact - split| of the threshold: the reported FPR95 is then the exact arm's, image for image.

Under `world_size > 1` (`Rescorer`) every rank re-scores only the window images of ITS shard of the set and the patches
ride one all-reduce of len(window) floats: no rank does work proportional to the whole window.

AUROC / AUPR are untouched in any digit that matters (they were within 1e-5 already); FPR@recall becomes the exact arm's.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple


def _threshold_interval(id_scores, recall: float):
    """The score interval the FPR@recall count depends on.  Scores are negated confidences (lower = more ID).  The reference
    (utils/detection_util.py:66-105) walks the thresholds in confidence order and takes the one whose recall is CLOSEST to
    `recall`; between two consecutive ID scores the recall does not change, and its argmin over the reversed arrays picks the
    far end of that run: with k = round(recall n) the count is #{OOD score < s_(k+1)}, s_(j) the j-th lowest ID score — not
    s_(k).  Which neighbour it is also depends on how recall n rounds, so the interval returned is [s_(k-1), s_(k+1)]: every
    image whose side of ANY of the three could change is inside the windows built around it (the exact operating point,
    closest recall and tie rules, is `mcm_measures`' business).  Round 4 centred the window on s_(k) alone, which was only
    right because its width (2.5 x the fp16 noise) dwarfed the spacing of neighbouring order statistics; the inner window
    of the two-level form (a few fp32 ulps) does not."""
    import torch

    n = id_scores.numel()
    k = min(n, max(1, int(round(recall * n))))
    srt = torch.sort(id_scores.double()).values
    return float(srt[max(0, k - 2)]), float(srt[min(n - 1, k)])   # 0-based: s_(k-1) = srt[k-2], s_(k+1) = srt[k]


def _calibration_indices(n: int, n_cal: int, device):
    """`n_cal` calibration images as runs of 64 consecutive indices spread evenly over the set: under `world_size > 1` the
    contiguous index shards then each hold about n_cal / W of them (the first n_cal images would all sit in rank 0's shard —
    the one stage of the refinement whose cost did not divide by W), and a loader that generates or decodes in blocks
    (`DevicePatternLoader.gather`: aligned 64-image blocks) still touches n_cal / 64 blocks only."""
    import torch

    if n_cal >= n:
        return torch.arange(n, device=device)
    run = 64
    blocks = -(-n // run)                       # aligned 64-image blocks of the set (the last one may be short)
    nb = min(blocks, -(-n_cal // run))
    picked = [(k * blocks) // nb for k in range(nb)]   # nb DISTINCT blocks, evenly spread (blocks >= nb): no image twice
    idx = torch.cat([torch.arange(b * run, min((b + 1) * run, n)) for b in picked])[:n_cal]
    return idx.to(device)   # (idx.numel() can fall short of n_cal by the short last block: callers report idx.numel())


class ThresholdRefiner:
    """Steps 1 - 2 on the ID set (`fit_id`), then step 3 on each OOD set as it arrives (`apply`): the CLI scores its OOD
    sets one after the other.  rescore(name, idx) -> better scores of images `idx` (LongTensor on the scores' device) of
    set `name` ("id" for the ID set); rescore_exact: the optional second level (module docstring)."""

    def __init__(self, rescore: Callable, *, rescore_exact: Optional[Callable] = None, recall: float = 0.95, margin: float = 2.5,
                 calib: int = 512, calib_exact: int = 16, max_rounds: int = 4):
        self.rescore, self.rescore_exact = rescore, rescore_exact
        self.recall, self.margin, self.calib, self.calib_exact, self.max_rounds = recall, margin, calib, calib_exact, max_rounds
        self.stats = {"recall": recall, "margin": margin, "rescored": {}, "rounds": 0}
        if rescore_exact is not None:
            self.stats["rescored_exact"] = {}
        self.delta, self.delta2, self.threshold, self.interval = None, None, None, None

    @staticmethod
    def _near(scores, iv, delta):
        return (scores >= iv[0] - delta) & (scores <= iv[1] + delta)

    def _window_rounds(self, scores, done, fn, delta, iv):
        """Re-score the not-yet-done images within `delta` of the threshold interval with `fn`, recompute the interval, repeat
        while it moves."""
        rounds = 0
        for r in range(self.max_rounds):
            rounds = r + 1
            idx = (self._near(scores, iv, delta) & ~done).nonzero().reshape(-1)
            if idx.numel():
                scores[idx] = fn("id", idx).to(device=scores.device, dtype=scores.dtype)
                done[idx] = True
            new = _threshold_interval(scores, self.recall)
            moved, iv = max(abs(new[0] - iv[0]), abs(new[1] - iv[1])), new
            if moved <= 0.5 * delta:
                break
        return iv, rounds

    def fit_id(self, id_scores):
        """Patches `id_scores` in place; afterwards `threshold` / `delta` (/ `delta2`) are set."""
        import torch

        dev, st = id_scores.device, self.stats
        n_cal = min(int(self.calib), id_scores.numel())
        idx = _calibration_indices(id_scores.numel(), n_cal, dev)
        better = self.rescore("id", idx).to(device=dev, dtype=torch.float32)
        n_cal = int(idx.numel())
        noise = float((better - id_scores[idx]).abs().max())
        id_scores[idx] = better
        done = torch.zeros(id_scores.numel(), dtype=torch.bool, device=dev)
        done[idx] = True
        self.delta = self.margin * noise
        st.update(noise_max_abs=noise, delta=self.delta, calibration_images=n_cal)
        iv = _threshold_interval(id_scores, self.recall)
        if self.delta > 0.0:  # (0: the arm IS the better arm)
            iv, st["rounds"] = self._window_rounds(id_scores, done, self.rescore, self.delta, iv)
        st["rescored"]["id"] = int(done.sum())
        if self.rescore_exact is not None:
            n2 = max(1, min(int(self.calib_exact), n_cal))
            idx = idx[:: max(1, n_cal // n2)][:n2]   # a spread subset of the (already split-arm-scored) calibration images
            exact = self.rescore_exact("id", idx).to(device=dev, dtype=torch.float32)
            # two exact-grade arms differ by a few fp32 ulps of the score; never taken below two ulps at the threshold
            a = max(abs(iv[0]), abs(iv[1]))
            ulp = float(torch.nextafter(torch.tensor(a, dtype=torch.float32), torch.tensor(float("inf"))) - a)
            noise2 = max(float((exact - id_scores[idx]).abs().max()), 2.0 * ulp)
            id_scores[idx] = exact
            done2 = torch.zeros(id_scores.numel(), dtype=torch.bool, device=dev)
            done2[idx] = True
            self.delta2 = self.margin * noise2
            st.update(noise2_max_abs=noise2, delta2=self.delta2, calibration_images_exact=n2)
            iv, st["rounds_exact"] = self._window_rounds(id_scores, done2, self.rescore_exact, self.delta2,
                                                         _threshold_interval(id_scores, self.recall))
            st["rescored_exact"]["id"] = int(done2.sum())
        self.interval = iv
        self.threshold = st["threshold"] = iv[1]   # (s_(k+1): the operating threshold when recall n is an integer)
        st["threshold_interval"] = list(iv)
        return id_scores

    def apply(self, name: str, scores):
        """Patches the scores of OOD set `name` in place."""
        assert self.threshold is not None, "fit_id first"
        n = 0
        if self.delta > 0.0:
            idx = self._near(scores, self.interval, self.delta).nonzero().reshape(-1)
            n = int(idx.numel())
            if n:
                scores[idx] = self.rescore(name, idx).to(device=scores.device, dtype=scores.dtype)
        self.stats["rescored"][name] = n
        self.stats["rescored_total"] = sum(self.stats["rescored"].values())
        if self.rescore_exact is not None:
            idx = self._near(scores, self.interval, self.delta2).nonzero().reshape(-1)
            if idx.numel():
                scores[idx] = self.rescore_exact(name, idx).to(device=scores.device, dtype=scores.dtype)
            self.stats["rescored_exact"][name] = int(idx.numel())
            self.stats["rescored_exact_total"] = sum(self.stats["rescored_exact"].values())
        return scores


def refine_threshold_scores(id_scores, ood_scores: Dict[str, "object"], rescore: Callable, *, rescore_exact: Optional[Callable] = None,
                            recall: float = 0.95, margin: float = 2.5, calib: int = 512, max_rounds: int = 4) -> Tuple[object, Dict, Dict]:
    """id_scores [n_id], ood_scores {name: [n]} — fp32 tensors of one 16-bit arm, patched IN PLACE and returned with the
    refiner's statistics."""
    r = ThresholdRefiner(rescore, rescore_exact=rescore_exact, recall=recall, margin=margin, calib=calib, max_rounds=max_rounds)
    r.fit_id(id_scores)
    for name, s in ood_scores.items():
        r.apply(name, s)
    return id_scores, ood_scores, r.stats


class Rescorer:
    """`rescore(name, idx)` over loaders: gathers the pixels of the named images (`loader.gather(idx)`, or `dataset[i]` of a
    map-style dataset) and scores them with `scorer` — anything with `score_images(pixels, bank, T, score)` and `max_batch`:
    `NativeCLIP.x2_scorer()` (the split-activation arm of the scoring handle) or an exact-fp32 `NativeCLIP` — against the
    same prompt bank.

    world_size > 1: every rank is called with the same `idx` (the gathered scores are identical on every rank); each
    re-scores only the images of ITS contiguous shard of the set (mcm_amd.dist.shard_range: the samples its loader shard
    decoded in the first place) and one all-reduce of len(idx) floats hands every rank every patch."""

    def __init__(self, scorer, bank, loaders: Dict[str, object], T: float, score: str):
        self.net, self.bank, self.loaders, self.T, self.score = scorer, bank, loaders, float(T), score
        self.scored_here = 0   # images this rank re-scored itself (sums to the window sizes over the ranks)

    def _score(self, name: str, ids):
        import torch

        loader = self.loaders[name]
        out = []
        bs = self.net.max_batch
        for s in range(0, len(ids), bs):
            chunk = ids[s:s + bs]
            if hasattr(loader, "gather"):
                px = loader.gather(chunk)
            else:  # reference-style DataLoader over a map-style dataset: item i is (image, label)
                px = torch.stack([loader.dataset[i][0] for i in chunk])
            out.append(self.net.score_images(px, self.bank, self.T, self.score))
        self.scored_here += len(ids)
        return torch.cat(out) if out else torch.empty(0, device=self.bank.device)

    def __call__(self, name: str, idx):
        import torch

        from . import dist as mdist

        rank, ws = mdist.world()
        if ws <= 1:
            return self._score(name, idx.tolist())
        lo, hi = mdist.shard_range(len(self.loaders[name].dataset), rank, ws)
        mine = (idx >= lo) & (idx < hi)
        out = torch.zeros(idx.numel(), dtype=torch.float32, device=self.bank.device)
        if bool(mine.any()):
            out[mine.to(out.device)] = self._score(name, idx[mine].tolist()).to(torch.float32)
        return mdist.all_reduce_sum(out)  # every entry has one non-zero contribution: x + 0 + ... + 0 is exact
