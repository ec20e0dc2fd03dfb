"""Resize(224)+CenterCrop(224): the C oracle vs outputs of Pillow itself (tests/golden/preprocess.npz,
written by tests/golden/make_golden.py with the reference's transform arithmetic)."""
import hashlib
import os

import numpy as np


def preprocess_input(h, w):
    """Same seeded input as tests/golden/make_golden.py::preprocess_input (only outputs are stored)."""
    return np.random.default_rng(h * 10007 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    return g, [tuple(int(v) for v in c) for c in g["cases"]]


def test_oracle_matches_pillow_bit_for_bit(golden_dir):
    from oracle import oracle as orc

    g, cases = _cases(golden_dir)
    assert len(cases) >= 8
    for h, w in cases:
        out = orc.resize_crop_u8(preprocess_input(h, w), 224)
        np.testing.assert_array_equal(out[:24, :24], g[f"patch_{h}x{w}"], err_msg=f"{h}x{w} corner")
        np.testing.assert_array_equal(out.astype(np.int64).sum(axis=(1, 2)), g[f"rowsum_{h}x{w}"])
        digest = np.frombuffer(hashlib.sha256(out.tobytes()).digest(), dtype=np.uint8)
        np.testing.assert_array_equal(digest, g[f"sha256_{h}x{w}"], err_msg=f"{h}x{w}")


def test_resized_size_rules():
    """torchvision's Resize(int): short side -> S, long side int(S*long/short), untouched if short == S."""
    import ctypes

    from oracle import oracle as orc

    def size(h, w, s=224):
        nh, nw = ctypes.c_int32(), ctypes.c_int32()
        orc.lib().orc_resized_size(h, w, s, ctypes.byref(nh), ctypes.byref(nw))
        return nh.value, nw.value

    assert size(375, 500) == (224, 298)
    assert size(500, 333) == (336, 224)
    assert size(224, 300) == (224, 300)
    assert size(64, 48) == (298, 224)
    assert size(231, 517) == (224, 501)


def test_identity_and_crop_only():
    from oracle import oracle as orc

    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    np.testing.assert_array_equal(orc.resize_crop_u8(img), img)
    wide = rng.integers(0, 256, (224, 301, 3), dtype=np.uint8)   # (301-224)/2 = 38.5 -> round-half-even 38
    np.testing.assert_array_equal(orc.resize_crop_u8(wide), wide[:, 38:38 + 224])
    tall = rng.integers(0, 256, (299, 224, 3), dtype=np.uint8)   # 37.5 -> 38
    np.testing.assert_array_equal(orc.resize_crop_u8(tall), tall[38:38 + 224])
