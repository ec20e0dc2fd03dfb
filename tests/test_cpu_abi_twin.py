"""The CPU twin of the C ABI (oracle/libmcm_cpu.so) and BASELINE config 1 ("CPU path, plumbing, no
GPU"): the SAME ctypes declarations the product binding uses (mcm_amd/config.py::CConfig, include/mcm.h's signatures)
drive the hot path on host pointers — set weights by HF name (fp32 and fp16 hand-over), encode the bank once, score
batches, the reference's metrics on the host — and the results equal the oracle's own front end.  Test infrastructure:
nothing under mcm_amd/ loads this library (asserted)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libmcm_cpu.so")
SECTION_8B = ["mcm_create", "mcm_set_weight", "mcm_encode_text", "mcm_encode_image", "mcm_score", "mcm_last_error",
              "mcm_destroy", "mcm_abi_version"]  # "what a C-ABI replacement must export"


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libmcm_cpu.so"], check=True, capture_output=True)
    L = ctypes.CDLL(LIB)
    from mcm_amd.config import CConfig

    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    L.mcm_create.argtypes = [ctypes.POINTER(CConfig), ctypes.POINTER(vp)]
    L.mcm_destroy.argtypes = [vp]
    L.mcm_destroy.restype = None
    L.mcm_last_error.argtypes = [vp]
    L.mcm_last_error.restype = ctypes.c_char_p
    L.mcm_set_weight.argtypes = [vp, ctypes.c_char_p, vp, i32, ctypes.POINTER(ctypes.c_int64), i32]
    L.mcm_finalize_weights.argtypes = [vp]
    L.mcm_encode_text.argtypes = [vp, vp, i32, i32, vp, vp]
    L.mcm_encode_image.argtypes = [vp, vp, i32, vp, vp]
    L.mcm_score.argtypes = [vp, vp, i32, vp, i32, f32, i32, vp, vp]
    L.mcm_weights_operand_exact.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(i32)]
    return L


def test_the_twin_exports_what_section_8b_lists_and_so_does_the_product_library(lib):
    from mcm_amd.config import ABI_VERSION
    from mcm_amd.engine import EXPORTED_SYMBOLS

    for sym in SECTION_8B:
        getattr(lib, sym)
        assert sym in EXPORTED_SYMBOLS
    assert lib.mcm_abi_version() == ABI_VERSION


def test_nothing_in_the_product_package_knows_the_twin():
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "mcm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libmcm_cpu" not in text and "mcm_cpu.c" not in text, os.path.join(dirpath, f)


@pytest.mark.parametrize("fp16_handover", [False, True])
def test_config1_plumbing_on_the_cpu(lib, fp16_handover):
    """ImageNet-10 ID vs ImageNet-20 OOD plumbing (K = 10 concepts) on the tiny geometry: weights -> bank -> batches of
    scores -> AUROC / FPR95, through the C ABI on the CPU; equal to the oracle front end, batch split invariant."""
    from mcm_amd.config import SCORE_KINDS, geometry
    from mcm_amd.metrics import get_measures
    from mcm_amd.synth import make_pixels, make_token_ids
    from mcm_amd.weights import synth_state_dict
    from oracle import oracle as orc

    geo = geometry("tiny")
    sd = synth_state_dict(geo, 0, "fp16-exact")
    K, n_id, n_ood, bs = 10, 48, 40, 16
    ids, _ = make_token_ids(K, seed=2)
    h = ctypes.c_void_p()
    cfg = geo.to_c(device=0, precision=1, max_batch=bs, max_prompt_tokens=K * ids.shape[1])
    assert lib.mcm_create(ctypes.byref(cfg), ctypes.byref(h)) == 0, lib.mcm_last_error(None)
    try:
        for name, arr in sd.items():
            a = np.ascontiguousarray(arr.astype(np.float16) if fp16_handover else arr)
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            assert lib.mcm_set_weight(h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), 1 if fp16_handover else 0,
                                      shape, a.ndim) == 0
        assert lib.mcm_finalize_weights(h) == 0
        n, sp = ctypes.c_uint64(9), ctypes.c_int32(9)
        assert lib.mcm_weights_operand_exact(h, ctypes.byref(n), ctypes.byref(sp)) == 0 and (n.value, sp.value) == (0, 0)
        ids32 = np.ascontiguousarray(ids, dtype=np.int32)
        bank = np.empty((K, geo.proj_dim), dtype=np.float32)
        assert lib.mcm_encode_text(h, ids32.ctypes.data_as(ctypes.c_void_p), K, ids.shape[1],
                                   bank.ctypes.data_as(ctypes.c_void_p), None) == 0

        def score_set(n_img, ood, seed):
            out = np.empty(n_img, dtype=np.float32)
            for s in range(0, n_img, bs):
                px, _ = make_pixels(min(bs, n_img - s), geo.image_size, K, ood=ood, seed=seed, start=s)
                b = px.shape[0]
                assert lib.mcm_score(h, px.ctypes.data_as(ctypes.c_void_p), b, bank.ctypes.data_as(ctypes.c_void_p), K, 1.0,
                                     SCORE_KINDS["MCM"], out[s:].ctypes.data_as(ctypes.c_void_p), None) == 0
            return out

        s_id, s_ood = score_set(n_id, False, 1), score_set(n_ood, True, 11)
        # the same through the oracle's own front end
        o = orc.OracleCLIP(geo, sd)
        t = o.encode_text(ids)
        px_id, _ = make_pixels(n_id, geo.image_size, K, ood=False, seed=1)
        px_ood, _ = make_pixels(n_ood, geo.image_size, K, ood=True, seed=11)
        np.testing.assert_array_equal(s_id, orc.score_features(o.encode_image(px_id), t, 1.0, 0))
        np.testing.assert_array_equal(s_ood, orc.score_features(o.encode_image(px_ood), t, 1.0, 0))
        auroc, aupr, fpr = get_measures(-s_id, -s_ood)
        assert 0.0 <= auroc <= 1.0 and 0.0 <= fpr <= 1.0 and np.isfinite(aupr)
        # error behaviour of the boundary: batch above max_batch, sequence above max_positions
        big, _ = make_pixels(bs + 1, geo.image_size, K, ood=False, seed=1)
        out = np.empty(bs + 1, dtype=np.float32)
        assert lib.mcm_score(h, big.ctypes.data_as(ctypes.c_void_p), bs + 1, bank.ctypes.data_as(ctypes.c_void_p), K, 1.0, 0,
                             out.ctypes.data_as(ctypes.c_void_p), None) == -7   # MCM_ERANGE
        assert b"max_batch" in lib.mcm_last_error(h)
        long_ids = np.full((1, 78), 49407, dtype=np.int32)
        assert lib.mcm_encode_text(h, long_ids.ctypes.data_as(ctypes.c_void_p), 1, 78, bank.ctypes.data_as(ctypes.c_void_p),
                                   None) == -7
    finally:
        lib.mcm_destroy(h)
