"""Threshold refinement: FPR@recall of a 16-bit arm made equal to the exact-fp32 arm's, at the cost of re-scoring a few
hundred images.

FPR95 (reference utils/detection_util.py:66-105) is a count: the OOD images on the ID side of ONE threshold, the score
below which 95 % of the ID set lies.  A 16-bit arm's score noise (rms 5.6e-9 for fp16 at B/16) is far smaller than the
scores' spread, so it can only change the count through the images whose score is within a few noise widths of that
threshold: 0 - 2 of 10 000 on the headline sets, 8 of 31 000 at a realistic operating point —
small, but not the reference's number.  Those images can be named: everything within `delta` of the threshold.  This
module re-scores exactly them with the exact-fp32 arm (pinned to HF at 1e-10 in score) and patches their scores in place;
every other image is provably on the same side of the threshold in both arms as long as its own noise is below `delta`.

    delta = margin x (largest |fp32 - 16-bit| score difference over a calibration sample of the ID set)

  1. calibration: the first `calib` ID images are re-scored -> noise estimate, delta;
  2. ID window: the ID images within delta of the provisional threshold are re-scored and the threshold recomputed
     (repeated if it moved by more than delta / 2);
  3. OOD windows: in every OOD set the images within delta of the final threshold are re-scored.

AUROC / AUPR are untouched in any digit that matters (they were within 1e-5 already); FPR@recall becomes the fp32 arm's.
The re-scoring callback is the caller's: the CLI hands in an fp32 `NativeCLIP` over the same loaders (`Rescorer`),
`mcm_amd.parity` the fp32 arm's scores it already holds.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple


def _kth_threshold(id_scores, recall: float):
    """Provisional threshold in score space.  Scores are negated confidences (lower = more ID): recall r of the ID set lies
    at or below the r-quantile.  The exact operating point (closest recall, tie rules) is `mcm_measures`' business; the
    window only has to contain it, and neighbouring order statistics are orders of magnitude closer than delta."""
    import torch

    n = id_scores.numel()
    k = min(n, max(1, int(round(recall * n))))
    return float(torch.kthvalue(id_scores.double(), k).values)


class ThresholdRefiner:
    """Steps 1 - 2 on the ID set (`fit_id`), then step 3 on each OOD set as it arrives (`apply`): the CLI scores its OOD
    sets one after the other.  rescore(name, idx) -> exact scores of images `idx` (LongTensor on the scores' device) of
    set `name` ("id" for the ID set)."""

    def __init__(self, rescore: Callable, *, recall: float = 0.95, margin: float = 2.5, calib: int = 512, max_rounds: int = 4):
        self.rescore, self.recall, self.margin, self.calib, self.max_rounds = rescore, recall, margin, calib, max_rounds
        self.stats = {"recall": recall, "margin": margin, "rescored": {}, "rounds": 0}
        self.delta, self.threshold = None, None

    def fit_id(self, id_scores):
        """Patches `id_scores` in place; afterwards `threshold` / `delta` are set."""
        import torch

        dev, st = id_scores.device, self.stats
        n_cal = min(int(self.calib), id_scores.numel())
        idx = torch.arange(n_cal, device=dev)
        exact = self.rescore("id", idx).to(device=dev, dtype=torch.float32)
        noise = float((exact - id_scores[idx]).abs().max())
        id_scores[idx] = exact
        done = torch.zeros(id_scores.numel(), dtype=torch.bool, device=dev)
        done[idx] = True
        self.delta = self.margin * noise
        st.update(noise_max_abs=noise, delta=self.delta, calibration_images=n_cal)
        t = _kth_threshold(id_scores, self.recall)
        if self.delta > 0.0:  # (0: the arm IS the exact arm)
            for r in range(self.max_rounds):
                st["rounds"] = r + 1
                idx = (((id_scores - t).abs() <= self.delta) & ~done).nonzero().reshape(-1)
                if idx.numel():
                    id_scores[idx] = self.rescore("id", idx).to(device=dev, dtype=torch.float32)
                    done[idx] = True
                t_new = _kth_threshold(id_scores, self.recall)
                moved, t = abs(t_new - t), t_new
                if moved <= 0.5 * self.delta:
                    break
        self.threshold = st["threshold"] = t
        st["rescored"]["id"] = int(done.sum())
        return id_scores

    def apply(self, name: str, scores):
        """Patches the scores of OOD set `name` in place."""
        import torch

        assert self.threshold is not None, "fit_id first"
        n = 0
        if self.delta > 0.0:
            idx = ((scores - self.threshold).abs() <= self.delta).nonzero().reshape(-1)
            n = int(idx.numel())
            if n:
                scores[idx] = self.rescore(name, idx).to(device=scores.device, dtype=torch.float32)
        self.stats["rescored"][name] = n
        self.stats["rescored_total"] = sum(self.stats["rescored"].values())
        return scores


def refine_threshold_scores(id_scores, ood_scores: Dict[str, "object"], rescore: Callable, *, recall: float = 0.95,
                            margin: float = 2.5, calib: int = 512, max_rounds: int = 4) -> Tuple[object, Dict, Dict]:
    """id_scores [n_id], ood_scores {name: [n]} — fp32 tensors of one 16-bit arm, patched IN PLACE and returned with the
    refiner's statistics."""
    r = ThresholdRefiner(rescore, recall=recall, margin=margin, calib=calib, max_rounds=max_rounds)
    r.fit_id(id_scores)
    for name, s in ood_scores.items():
        r.apply(name, s)
    return id_scores, ood_scores, r.stats


class Rescorer:
    """`rescore(name, idx)` over loaders: gathers the pixels of the named images (`loader.gather(idx)`, or `dataset[i]` of a
    map-style dataset) and scores them with the exact-fp32 handle against the same prompt bank."""

    def __init__(self, net32, bank, loaders: Dict[str, object], T: float, score: str):
        self.net, self.bank, self.loaders, self.T, self.score = net32, bank, loaders, float(T), score

    def __call__(self, name: str, idx):
        import torch

        loader = self.loaders[name]
        out = []
        ids = idx.tolist()
        bs = self.net.max_batch
        for s in range(0, len(ids), bs):
            chunk = ids[s:s + bs]
            if hasattr(loader, "gather"):
                px = loader.gather(chunk)
            else:  # reference-style DataLoader over a map-style dataset: item i is (image, label)
                px = torch.stack([loader.dataset[i][0] for i in chunk])
            out.append(self.net.score_images(px, self.bank, self.T, self.score))
        return torch.cat(out) if out else torch.empty(0, device=self.bank.device)
