"""CPU-side checks: the C-ABI library loads and exports every symbol include/mcm.h declares
(no compute calls without a GPU), the ctypes config mirrors the C struct, and the host logic
(tokenizer stand-in, loaders, sharding) behaves like the reference's contract."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

from mcm_amd import config as cfgmod
from mcm_amd.engine import EXPORTED_SYMBOLS, HARNESS_LIB_PATH, HARNESS_ONLY_SYMBOLS, LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return ctypes.CDLL(LIB_PATH)


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "mcm.h")).read()
    harness_part = re.search(r"#ifdef MCM_HARNESS(.*?)#endif", header, re.S).group(1)
    product_part = header.replace(harness_part, "")
    declared = set(re.findall(r"\b(mcm_[a-z0-9_]+)\s*\(", product_part))
    declared -= {"mcm_handle", "mcm_config"}
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    lib.mcm_abi_version.restype = ctypes.c_int32
    assert lib.mcm_abi_version() == cfgmod.ABI_VERSION
    # the A/B switches exist in the harness library only: the shipped one does not export them
    debug = set(re.findall(r"\b(mcm_[a-z0-9_]+)\s*\(", harness_part))
    assert debug == set(HARNESS_ONLY_SYMBOLS)
    for sym in debug:
        assert not hasattr(lib, sym), f"{sym} must not be exported by libmcm_hip.so"
    hlib = ctypes.CDLL(HARNESS_LIB_PATH)
    for sym in declared | debug:
        assert getattr(hlib, sym) is not None


def test_config_struct_matches_header():
    header = open(os.path.join(ROOT, "include", "mcm.h")).read()
    body = header[header.index("typedef struct mcm_config {"):header.index("} mcm_config;")]
    fields = re.findall(r"^\s*(?:int32_t|float)\s+(\w+);", body, flags=re.M)
    assert fields == [f[0] for f in cfgmod.CConfig._fields_]
    assert ctypes.sizeof(cfgmod.CConfig) == 4 * len(fields)


def test_create_rejects_bad_config_without_gpu(lib):
    lib.mcm_create.argtypes = [ctypes.POINTER(cfgmod.CConfig), ctypes.POINTER(ctypes.c_void_p)]
    lib.mcm_last_error.restype = ctypes.c_char_p
    lib.mcm_last_error.argtypes = [ctypes.c_void_p]
    h = ctypes.c_void_p()
    bad = cfgmod.geometry("ViT-B/16").to_c()
    bad.abi_version = 99
    assert lib.mcm_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert b"ABI" in lib.mcm_last_error(None)
    bad = cfgmod.geometry("ViT-B/16").to_c()
    bad.v_heads = 7  # head_dim != 64
    assert lib.mcm_create(ctypes.byref(bad), ctypes.byref(h)) == -1


def test_engine_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mcm_amd.engine import NativeCLIP

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NativeCLIP("tiny", {})


def test_geometry_flops_match_survey():
    assert abs(cfgmod.geometry("ViT-B/16").vision_flops_per_image() / 1e9 - 35.13) < 0.02
    assert abs(cfgmod.geometry("ViT-B/32").vision_flops_per_image() / 1e9 - 8.82) < 0.02
    assert abs(cfgmod.geometry("ViT-L/14").vision_flops_per_image() / 1e9 - 162.03) < 0.05


def test_param_schema_counts():
    from mcm_amd.weights import param_shapes

    n = sum(int(np.prod(s)) for s in param_shapes(cfgmod.geometry("ViT-B/16")).values())
    assert abs(n / 1e6 - 149.6) < 0.1  # SURVEY §2.1 (minus the scalar logit_scale)


def test_tokenizer_contract():
    from mcm_amd.tokenizer import BOS, EOS, HashTokenizer

    out = HashTokenizer()(["a photo of a tench", "a photo of a toilet paper"], padding=True,
                          return_tensors="np")
    ids, mask = out["input_ids"], out["attention_mask"]
    assert ids.shape == (2, 8) and (ids[:, 0] == BOS).all()
    assert ids[0, 6] == EOS and ids[0, 7] == EOS and mask[0].tolist() == [1] * 7 + [0]
    assert (ids.argmax(axis=1) == mask.sum(axis=1) - 1).all()  # EOS is the first max id
    assert (ids[0, 1:5] == ids[1, 1:5]).all()  # same words → same ids


def test_load_tokenizer_never_returns_degenerate_tokenizer():
    """No vocabulary exists offline; transformers 5.x then hands back an empty tokenizer that maps
    every prompt to the same ids — load_tokenizer must detect that and fall back."""
    import pytest

    from mcm_amd.tokenizer import BOS, EOS, TokenizerUnavailable, load_tokenizer

    with pytest.warns(RuntimeWarning, match="HashTokenizer stand-in"):   # never a silent fallback
        tok = load_tokenizer("openai/clip-vit-base-patch16")
    with pytest.raises(TokenizerUnavailable):                            # real weights: refuse stand-in ids
        load_tokenizer("openai/clip-vit-base-patch16", allow_hash=False)
    ids = np.asarray(tok([f"a photo of a concept{k:04d}" for k in range(5)], padding=True,
                         return_tensors="np")["input_ids"])
    assert (ids[:, 0] == BOS).all() and ids.max() == EOS
    assert len({tuple(r) for r in ids.tolist()}) == 5


def test_synthetic_loader_is_shard_consistent():
    from mcm_amd.synth import SyntheticImageSet, SyntheticLoader

    ds = SyntheticImageSet(23, 32, 5, ood=False, seed=1)
    full = np.concatenate([x.numpy() for x, _ in SyntheticLoader(ds, 8)])
    a = np.concatenate([x.numpy() for x, _ in SyntheticLoader(ds, 5).shard(0, 12)])
    b = np.concatenate([x.numpy() for x, _ in SyntheticLoader(ds, 7).shard(12, 23)])
    assert full.shape == (23, 3, 32, 32) and np.array_equal(full, np.concatenate([a, b]))


def test_shard_ranges_cover_in_order():
    from mcm_amd.dist import shard_range

    for n in (0, 1, 7, 500, 50000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_get_ood_scores_clip_contract_with_stub_net():
    """Host logic only: batching, concat, truncation, dtype, prompt hoisting — with a stub
    `net` (numpy arithmetic standing in for the device kernels)."""
    import torch

    from mcm_amd.detection import get_ood_scores_clip
    from mcm_amd.synth import SyntheticImageSet, SyntheticLoader, class_names

    calls = {"text": 0, "img": 0}

    class Stub:
        def get_text_features(self, input_ids, attention_mask):
            calls["text"] += 1
            assert input_ids.shape[0] == 6
            return torch.eye(6, 8)

        def score_images(self, images, text, T, score):
            calls["img"] += 1
            return -images.reshape(images.shape[0], -1).mean(dim=1).float()

    ds = SyntheticImageSet(21, 16, 6, ood=False, seed=1)
    args = types.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1)
    s = get_ood_scores_clip(args, Stub(), SyntheticLoader(ds, 8), class_names(6), in_dist=True)
    assert s.dtype == np.float32 and s.shape == (21,) and calls == {"text": 1, "img": 3}
    with pytest.raises(TypeError):
        get_ood_scores_clip(args, object(), SyntheticLoader(ds, 8), class_names(6))
    args.score = "maha"
    with pytest.raises(ValueError):
        get_ood_scores_clip(args, Stub(), SyntheticLoader(ds, 8), class_names(6))


def test_native_image_packing_equals_a_python_copy():
    """mcm_pack_u8 (csrc/ingest.cpp, host code: runs without a GPU): n images of their own sizes into one buffer at given
    offsets, the work cut by bytes over native threads — same bytes as a Python loop, untouched gaps, range errors."""
    import ctypes

    import numpy as np

    from mcm_amd.engine import LIB_PATH

    L = ctypes.CDLL(LIB_PATH)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    L.mcm_pack_u8.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.c_int32, vp, i64, ctypes.c_int32]
    rng = np.random.default_rng(3)
    shapes = [(int(rng.integers(1, 400)), int(rng.integers(1, 500)), 3) for _ in range(97)] + [(1200, 1600, 3), (1, 1, 3)]
    imgs = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in shapes]
    offs, o = [], 0
    for a in imgs:
        offs.append(o)
        o += (a.size + 15) // 16 * 16
    for threads in (1, 3, 16, 64):
        dst = np.full(o + 32, 0xAB, dtype=np.uint8)
        want = dst.copy()
        for a, off in zip(imgs, offs):
            want[off:off + a.size] = a.reshape(-1)
        n = len(imgs)
        srcs = (vp * n)(*[a.ctypes.data for a in imgs])
        sizes = (i64 * n)(*[a.size for a in imgs])
        offsets = (i64 * n)(*offs)
        assert L.mcm_pack_u8(srcs, sizes, offsets, n, dst.ctypes.data_as(vp), dst.size, threads) == 0
        assert np.array_equal(dst, want), threads
    assert L.mcm_pack_u8(srcs, sizes, offsets, n, dst.ctypes.data_as(vp), o - 64, 4) == -7     # MCM_ERANGE: last image does not fit
    assert L.mcm_pack_u8(srcs, sizes, offsets, 0, dst.ctypes.data_as(vp), dst.size, 4) == 0


def test_full_round_batches():
    """ClipGeometry.full_round_batches / gemm_tile_rounds (EXPERIMENTS.md R5.11): the batches at which every vision GEMM
    fills its last round of the persistent 256-workgroup grid; batch 512 of the headline config does not."""
    from mcm_amd.config import geometry

    b16, l14, b32 = geometry("ViT-B/16"), geometry("ViT-L/14"), geometry("ViT-B/32")
    assert b16.full_round_batches(1, 1400) == [332, 665, 998, 1330]
    assert l14.full_round_batches(200, 520) == [255, 318, 382, 446, 510]
    assert b32.full_round_batches(1, 2048) == [1310]
    r = b16.gemm_tile_rounds(512)
    assert {k: v["rounds"] for k, v in r.items()} == {"qkv": 15, "outproj": 5, "fc1": 19, "fc2": 5}
    assert abs(r["outproj"]["fill"] - 394 * 3 / (5 * 256)) < 1e-12
    assert all(v["fill"] == 1.0 for v in b16.gemm_tile_rounds(665).values())
    assert l14.gemm_tile_rounds(256)["outproj"]["rounds"] == 5 and l14.gemm_tile_rounds(255)["outproj"]["rounds"] == 4
    for b in b16.full_round_batches(1, 1400):  # the largest batch of its M-tile count
        assert -(-(b + 1) * 197 // 256) == -(-b * 197 // 256) + 1
