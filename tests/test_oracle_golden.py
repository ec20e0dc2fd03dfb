"""Pin the CPU oracle (oracle/mcm_oracle.c) to the reference's arithmetic.

Fixtures: tests/golden/*.npz, captured by tests/golden/make_golden.py from HF
transformers CLIPModel (the library the reference delegates to) and from the reference's
own get_ood_scores_clip, on the seeded weights/inputs that mcm_amd regenerates here.
Tolerances are fp32 round-off for different summation orders (torch/oneDNN vs the C loops).
"""
import os

import numpy as np
import pytest

from mcm_amd.config import SCORE_KINDS, geometry
from mcm_amd.synth import make_pixels, make_token_ids
from mcm_amd.weights import synth_state_dict
from oracle import oracle as orc

ATOL, RTOL = 2e-4, 2e-4


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.fixture(scope="module")
def tiny():
    geo = geometry("tiny")
    return geo, orc.OracleCLIP(geo, synth_state_dict(geo, 0))


def test_tiny_vision_hidden_states(golden_dir, tiny):
    geo, o = tiny
    g = _load(golden_dir, "clip_tiny.npz")
    px, _ = make_pixels(int(g["n_img"]), geo.image_size, 10, ood=False, seed=1)
    for i in range(geo.v_layers + 1):
        got = o.vision_hidden(px, i)
        np.testing.assert_allclose(got, g[f"v_hidden_{i}"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o.encode_image(px, normalize=False), g["image_features"],
                               rtol=RTOL, atol=ATOL)


def test_tiny_text_hidden_states(golden_dir, tiny):
    geo, o = tiny
    g = _load(golden_dir, "clip_tiny.npz")
    ids, mask = make_token_ids(int(g["n_txt"]), seed=2)
    for i in range(geo.t_layers + 1):
        got = o.text_hidden(ids, i)
        want = g[f"t_hidden_{i}"]
        for k in range(ids.shape[0]):  # rows up to and including EOS (pads differ: HF masks them)
            n = int(mask[k].sum())
            np.testing.assert_allclose(got[k, :n], want[k, :n], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o.encode_text(ids, normalize=False), g["text_features"],
                               rtol=RTOL, atol=ATOL)
    assert float(g["text_features_nomask_maxdiff"]) == 0.0  # padding-invariance KAT


@pytest.mark.parametrize("name,fixture", [("B16-2L", "clip_B16-2L.npz"),
                                          ("ViT-B/16", "clip_ViT-B_16.npz"),
                                          ("ViT-B/32", "clip_ViT-B_32.npz"),
                                          ("ViT-L/14", "clip_ViT-L_14.npz")])
def test_full_width_towers(golden_dir, name, fixture):
    geo = geometry(name)
    o = orc.OracleCLIP(geo, synth_state_dict(geo, 0))
    g = _load(golden_dir, fixture)
    px, _ = make_pixels(int(g["n_img"]), geo.image_size, 10, ood=False, seed=1)
    rows = g["v_rows"]
    for i in sorted({0, 1, geo.v_layers}):
        got = o.vision_hidden(px, i)[:, rows, :]
        np.testing.assert_allclose(got, g[f"v_hidden_{i}"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(o.encode_image(px, normalize=False), g["image_features"],
                               rtol=5e-4, atol=5e-4)
    ids, _ = make_token_ids(int(g["n_txt"]), seed=2)
    got = o.text_hidden(ids, geo.t_layers)[:, :4, :]
    np.testing.assert_allclose(got, g[f"t_hidden_{geo.t_layers}"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(o.encode_text(ids, normalize=False), g["text_features"],
                               rtol=5e-4, atol=5e-4)


def test_reference_scores_all_kinds(golden_dir, tiny):
    """Oracle end-to-end vs the reference's own get_ood_scores_clip output."""
    geo, o = tiny
    g = _load(golden_dir, "scores_tiny.npz")
    K, n_id, n_ood = int(g["K"]), int(g["n_id"]), int(g["n_ood"])
    ids, _ = make_token_ids(K, seed=2)
    txt = o.encode_text(ids)
    px_in, _ = make_pixels(n_id, geo.image_size, K, ood=False, seed=1)
    px_out, _ = make_pixels(n_ood, geo.image_size, K, ood=True, seed=1)
    f_in, f_out = o.encode_image(px_in), o.encode_image(px_out)
    for score, kind in SCORE_KINDS.items():
        for T in (1, 2):
            s_in = orc.score_features(f_in, txt, T, kind)
            s_out = orc.score_features(f_out, txt, T, kind)
            tol = dict(rtol=2e-5, atol=1e-6) if score != "var" else dict(rtol=2e-3, atol=1e-9)
            np.testing.assert_allclose(s_in, g[f"{score}_T{T}_in"], **tol)
            np.testing.assert_allclose(s_out, g[f"{score}_T{T}_out"], **tol)
            assert s_in.dtype == np.float32


def test_reference_scores_measures_match(golden_dir, tiny):
    from mcm_amd.metrics import get_measures

    geo, o = tiny
    g = _load(golden_dir, "scores_tiny.npz")
    K = int(g["K"])
    ids, _ = make_token_ids(K, seed=2)
    txt = o.encode_text(ids)
    px_in, _ = make_pixels(int(g["n_id"]), geo.image_size, K, ood=False, seed=1)
    px_out, _ = make_pixels(int(g["n_ood"]), geo.image_size, K, ood=True, seed=1)
    for T in (1, 2):
        s_in = orc.score_features(o.encode_image(px_in), txt, T, 0)
        s_out = orc.score_features(o.encode_image(px_out), txt, T, 0)
        got = np.array(get_measures(-s_in, -s_out))
        np.testing.assert_allclose(got, g[f"measures_T{T}"], atol=1e-4)
