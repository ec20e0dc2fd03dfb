"""ctypes front-end of the CPU oracle (oracle/mcm_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by mcm_amd/.  Parity status: pinned against fixtures captured
from the reference's arithmetic (HF transformers CLIPModel, the reference's
get_ood_scores_clip / get_measures) by tests/golden/make_golden.py; checked in
tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmcm_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mcm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "libmcm_oracle.so"],
                       check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        f32p, i32p, i64p = (ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32),
                            ctypes.POINTER(ctypes.c_int64))
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        L.orc_create.argtypes = [vp, ctypes.POINTER(vp)]
        L.orc_destroy.argtypes = [vp]
        L.orc_destroy.restype = None
        L.orc_last_error.argtypes = [vp]
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_set_weight.argtypes = [vp, ctypes.c_char_p, f32p, i64p, i32]
        L.orc_layernorm.argtypes = [f32p, f32p, f32p, f32p, i64, i32, f32]
        L.orc_layernorm.restype = None
        L.orc_linear.argtypes = [f32p, f32p, f32p, f32p, i64, i32, i32]
        L.orc_linear.restype = None
        L.orc_quick_gelu.argtypes = [f32p, i64]
        L.orc_quick_gelu.restype = None
        L.orc_attention.argtypes = [f32p, f32p, i32, i32, i32, i32, i32]
        L.orc_attention.restype = None
        L.orc_l2_normalize.argtypes = [f32p, i64, i32]
        L.orc_l2_normalize.restype = None
        L.orc_vision_hidden.argtypes = [vp, f32p, i32, i32, f32p]
        L.orc_encode_image.argtypes = [vp, f32p, i32, f32p, i32]
        L.orc_text_hidden.argtypes = [vp, i32p, i32, i32, i32, f32p]
        L.orc_encode_text.argtypes = [vp, i32p, i32, i32, f32p, i32]
        L.orc_score_features.argtypes = [f32p, i32, f32p, i32, i32, f32, i32, f32p]
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.orc_resize_crop_u8.argtypes = [u8p, i32, i32, i32, u8p]
        L.orc_resized_size.argtypes = [i32, i32, i32, i32p, i32p]
        L.orc_resized_size.restype = None
        L.orc_maha_scores.argtypes = [f32p, i32, f32p, i32, f32p, i32, f32p]
        L.orc_maha_scores.restype = None
        _lib = L
    return _lib


def _f(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _c32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- operator level -------------------------------------------------------------------

def layernorm(x, g, b, eps=1e-5):
    x = _c32(x); g = _c32(g); b = _c32(b)
    y = np.empty_like(x)
    lib().orc_layernorm(_f(x), _f(g), _f(b), _f(y), x.size // x.shape[-1], x.shape[-1], eps)
    return y


def linear(x, w, bias=None):
    x = _c32(x); w = _c32(w)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    y = np.empty((M, N), dtype=np.float32)
    bp = _f(_c32(bias)) if bias is not None else None
    lib().orc_linear(_f(x), _f(w), bp, _f(y), M, N, K)
    return y


def quick_gelu(x):
    y = _c32(x).copy()
    lib().orc_quick_gelu(_f(y), y.size)
    return y


def attention(qkv, nseq, L, heads, hd=64, causal=False):
    qkv = _c32(qkv)
    assert qkv.shape == (nseq * L, 3 * heads * hd)
    out = np.empty((nseq * L, heads * hd), dtype=np.float32)
    lib().orc_attention(_f(qkv), _f(out), nseq, L, heads, hd, int(causal))
    return out


def score_features(img, text, T=1.0, kind=0):
    img = _c32(img); text = _c32(text)
    B, Pd = img.shape
    K = text.shape[0]
    out = np.empty(B, dtype=np.float32)
    rc = lib().orc_score_features(_f(img), B, _f(text), K, Pd, float(T), int(kind), _f(out))
    if rc:
        raise RuntimeError(f"orc_score_features rc={rc}")
    return out


def resize_crop_u8(img, size=224):
    """[H,W,3] uint8 RGB -> [size,size,3]: Resize(size) + CenterCrop(size) of the reference's loader."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, C = img.shape
    assert C == 3
    out = np.empty((size, size, 3), dtype=np.uint8)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    rc = lib().orc_resize_crop_u8(img.ctypes.data_as(u8p), H, W, size, out.ctypes.data_as(u8p))
    if rc:
        raise RuntimeError(f"orc_resize_crop_u8 rc={rc}")
    return out


def jpeg_reconstruct(coef, quant, width, height, hs, vs, wb, hb):
    """Quantised DCT coefficients -> RGB uint8 [height,width,3] as libjpeg's default decompressor gives (oracle/jpeg_ref.c).
    coef: list of int16 arrays [hb_c, wb_c, 64] (natural order), one per component; quant: uint16 [ncomp, 64]."""
    ncomp = len(coef)
    coef = [np.ascontiguousarray(c, dtype=np.int16) for c in coef]
    quant = np.ascontiguousarray(quant, dtype=np.uint16)
    out = np.empty((height, width, 3), dtype=np.uint8)
    ptrs = (ctypes.c_void_p * 3)(*([c.ctypes.data for c in coef] + [None] * (3 - ncomp)))
    i3 = lambda v: (ctypes.c_int * 3)(*(list(v) + [0] * (3 - len(v))))  # noqa: E731
    f = lib().orc_jpeg_reconstruct
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = f(ptrs, quant.ctypes.data, width, height, ncomp, i3(hs), i3(vs), i3(wb), i3(hb), out.ctypes.data)
    if rc:
        raise RuntimeError(f"orc_jpeg_reconstruct rc={rc}")
    return out


def maha_scores(feats, means, prec):
    """get_Mahalanobis_score's per-sample value for features [B,P], class means [C,P], precision [P,P]."""
    feats = _c32(feats); means = _c32(means); prec = _c32(prec)
    B, Pd = feats.shape
    out = np.empty(B, dtype=np.float32)
    lib().orc_maha_scores(_f(feats), B, _f(means), means.shape[0], _f(prec), Pd, _f(out))
    return out


# ---- model level ----------------------------------------------------------------------

class OracleCLIP:
    """CPU restatement of `net` (get_image_features / get_text_features)."""

    def __init__(self, geo, state_dict: Dict[str, np.ndarray]):
        self.geo = geo
        self._cfg = geo.to_c()
        self._h = ctypes.c_void_p()
        rc = lib().orc_create(ctypes.byref(self._cfg), ctypes.byref(self._h))
        if rc:
            raise RuntimeError(f"orc_create rc={rc}")
        for name, arr in state_dict.items():
            a = _c32(arr)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            rc = lib().orc_set_weight(self._h, name.encode(), _f(a), shape, a.ndim)
            if rc:
                raise RuntimeError(f"orc_set_weight({name}) rc={rc}")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def _check(self, rc):
        if rc:
            raise RuntimeError(f"oracle rc={rc}: {lib().orc_last_error(self._h).decode()}")

    def vision_hidden(self, pixels, nlayers):
        px = _c32(pixels)
        B = px.shape[0]
        out = np.empty((B, self.geo.v_tokens, self.geo.v_width), dtype=np.float32)
        self._check(lib().orc_vision_hidden(self._h, _f(px), B, nlayers, _f(out)))
        return out

    def encode_image(self, pixels, normalize=True):
        px = _c32(pixels)
        B = px.shape[0]
        out = np.empty((B, self.geo.proj_dim), dtype=np.float32)
        self._check(lib().orc_encode_image(self._h, _f(px), B, _f(out), int(normalize)))
        return out

    def text_hidden(self, ids, nlayers):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        K, S = ids.shape
        out = np.empty((K, S, self.geo.t_width), dtype=np.float32)
        self._check(lib().orc_text_hidden(
            self._h, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), K, S, nlayers, _f(out)))
        return out

    def encode_text(self, ids, normalize=True):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        K, S = ids.shape
        out = np.empty((K, self.geo.proj_dim), dtype=np.float32)
        self._check(lib().orc_encode_text(
            self._h, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), K, S, _f(out),
            int(normalize)))
        return out

    def scores(self, pixels, text_feats, T=1.0, kind=0):
        return score_features(self.encode_image(pixels), text_feats, T, kind)
