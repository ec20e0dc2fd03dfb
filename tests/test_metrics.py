"""mcm_amd.metrics vs the reference's get_measures outputs (tests/golden/measures.npz)."""
import os

import numpy as np
import pytest

from mcm_amd.metrics import fpr_at_recall, get_measures


@pytest.mark.parametrize("case", ["kat", "gauss", "ties", "narrow", "equal", "sep"])
def test_get_measures_matches_reference(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "measures.npz"))
    got = np.array(get_measures(g[f"{case}_pos"], g[f"{case}_neg"]), dtype=np.float64)
    np.testing.assert_allclose(got, g[f"{case}_measures"], rtol=0, atol=1e-12)


def test_known_answer():
    auroc, aupr, fpr = get_measures([.9, .8, .7, .6], [.65, .5, .4, .3, .2])
    assert (round(auroc, 12), round(aupr, 12), round(fpr, 12)) == (0.95, 0.95, 0.2)


def test_fpr_is_order_invariant():
    rng = np.random.default_rng(0)
    s = rng.normal(size=300).astype(np.float32)
    y = rng.random(300) < 0.4
    p = rng.permutation(300)
    assert fpr_at_recall(y, s) == fpr_at_recall(y[p], s[p])
