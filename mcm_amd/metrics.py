"""OOD-detection metrics of the reference: AUROC, AUPR, FPR at 95 % recall.

Re-statement of `get_measures` / `fpr_and_fdr_at_recall` (reference
utils/detection_util.py:66-119), written from scratch and pinned to the reference's
outputs by tests/golden/measures.npz (incl. heavy ties, all-equal and separated cases).
Host-side by design: O(N log N) on ≤60k floats (<10 ms), and it defines what
"AUROC/FPR95 match" means, so it stays in float64 numpy exactly like the reference.
ID samples are the positive class (reference :112-113).
"""
from __future__ import annotations

import numpy as np


def fpr_at_recall(is_pos: np.ndarray, score: np.ndarray, recall_level: float = 0.95) -> float:
    """False-positive rate at the operating point whose recall is closest to
    `recall_level`, scanning operating points from full recall downwards and keeping the
    first closest one (the tie rule the reference's reversed-slice argmin implies,
    utils/detection_util.py:100-106)."""
    is_pos = np.asarray(is_pos, dtype=bool).ravel()
    score = np.asarray(score).ravel()
    order = np.argsort(score, kind="stable")[::-1]
    s = score[order]
    y = is_pos[order]
    # one operating point per distinct score value: the last index of each run
    ends = np.flatnonzero(s[1:] != s[:-1])
    ends = np.append(ends, s.size - 1)
    tp = np.cumsum(y, dtype=np.float64)[ends]
    fp = (ends + 1) - tp
    n_pos = tp[-1]
    recall = tp / n_pos
    # operating points past the first one that reaches full recall add nothing
    full = int(np.searchsorted(tp, n_pos))
    cand_recall = recall[full::-1]
    cand_fp = fp[full::-1]
    pick = int(np.argmin(np.abs(cand_recall - recall_level)))
    return float(cand_fp[pick] / np.count_nonzero(~is_pos))


def get_measures(_pos, _neg, recall_level: float = 0.95):
    """(auroc, aupr, fpr) for ID scores `_pos` vs OOD scores `_neg`
    (same signature as reference utils/detection_util.py:108)."""
    import sklearn.metrics as sk  # the reference's own metric backend (:7,:115-116)

    pos = np.asarray(_pos).reshape(-1)
    neg = np.asarray(_neg).reshape(-1)
    examples = np.concatenate([pos, neg])
    labels = np.zeros(examples.size, dtype=np.int32)
    labels[:pos.size] = 1
    auroc = sk.roc_auc_score(labels, examples)
    aupr = sk.average_precision_score(labels, examples)
    fpr = fpr_at_recall(labels == 1, examples, recall_level)
    return auroc, aupr, fpr
