"""Mahalanobis baseline: the C oracle's scoring loop vs the reference's own get_Mahalanobis_score
outputs (tests/golden/maha_tiny.npz, captured by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest


@pytest.mark.parametrize("tag", ["raw", "norm"])
def test_oracle_scores_match_reference(golden_dir, tag):
    from oracle import oracle as orc

    g = np.load(os.path.join(golden_dir, "maha_tiny.npz"))
    got = orc.maha_scores(g[f"feat_in_{tag}"], g[f"mean_{tag}"], g[f"prec_{tag}"])
    np.testing.assert_allclose(got, g[f"in_{tag}"], rtol=2e-5, atol=1e-5)


def test_reference_drops_the_trailing_ood_batch(golden_dir):
    g = np.load(os.path.join(golden_dir, "maha_tiny.npz"))
    n_ood, bs = int(g["n_ood"]), int(g["batch"])
    assert g["out_raw"].shape[0] == (n_ood // bs) * bs < n_ood   # utils/detection_util.py:185-186
    assert g["in_raw"].shape[0] == int(g["n_id"])
