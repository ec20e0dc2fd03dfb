"""The split-activation arm (include/mcm.h mcm_score_x2) on a real MI355X: every kernel that carries an
activation as hi + lo fp16 pairs against the CPU oracle through the operator-level entry points, then the whole tower against
the exact-fp32 arm of the same library.  The bar is fp32 round-off, not fp16's: a split operand carries ~22 significand bits.
Reference arithmetic: HF modeling_clip.py:298-335 (attention), :346-350 (MLP), :362-383 (layer); the reference's tail
utils/detection_util.py:225-248."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

F16 = 2  # MCM_PREC_F16
SPLIT_W, SPLIT_X, SPLIT_OUT = 1, 2, 4


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def split_image(x: np.ndarray) -> np.ndarray:
    """fp32 [M, K] (K % 64 == 0) -> the split image [M, 2K] as float16: per 64 columns hi[64] then lo[64]."""
    M, K = x.shape
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    out = np.empty((M, K // 64, 2, 64), np.float16)
    out[:, :, 0, :] = hi.reshape(M, K // 64, 64)
    out[:, :, 1, :] = lo.reshape(M, K // 64, 64)
    return out.reshape(M, 2 * K)


def merge_image(y: np.ndarray) -> np.ndarray:
    """split image [M, 2N] float16 -> fp32 [M, N] = hi + lo (exact in fp64, returned as float64)."""
    M, N2 = y.shape
    v = y.reshape(M, N2 // 128, 2, 64).astype(np.float64)
    return (v[:, :, 0, :] + v[:, :, 1, :]).reshape(M, N2 // 2)


@pytest.fixture(scope="module")
def net():
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry("tiny")
    n = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=64, max_prompt_tokens=4096)
    yield n
    n.close()


def test_split_image_helpers_round_trip():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((5, 128)) * 3).astype(np.float32)
    back = merge_image(split_image(x))
    assert np.abs(back - x).max() <= 2.0 ** -21 * np.abs(x).max()   # ~22 bits (11 + 11)


X2_SHAPES = [
    (128, 128, 64),        # one tile, one logical K-step (two passes)
    (300, 256, 128),       # ragged M, tile kernels
    (50, 192, 128),        # 64x128 tiles, masked columns
    (2600, 768, 192),      # persistent kernel with edge tiles (gemm_p256)
    (4096, 768, 768),      # whole tiles: the ping-pong kernel (out-proj shape)
    (2048, 3072, 768),     # fc1 shape, ping-pong
    (1024, 768, 3072),     # fc2 shape, long K
]


@pytest.mark.parametrize("M,N,K", X2_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_linear_with_a_split_activation(net, M, N, K, epi):
    """y = (X_hi + X_lo) W^T + b against the float64 product of the SAME operands (X to 22 bits, W an exact fp16 number):
    fp32 accumulation is the only error left; the 16-bit outputs (epilogues 0 / 1) then round to fp16."""
    rng = np.random.default_rng(M + N + K + epi)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    b = (0.1 * rng.standard_normal(N)).astype(np.float32)
    xs = split_image(x)
    want = merge_image(xs) @ w.astype(np.float64).T + b
    xd, wd, bd = _dev(xs), _dev(w), _dev(b)
    lib, h = net._lib, net._h
    if epi == 2:
        r0 = rng.standard_normal((M, N)).astype(np.float32)
        rd = _dev(r0)
        rc = lib.mcm_op_linear_ex(h, F16, _ptr(xd), _ptr(wd), _ptr(bd), None, _ptr(rd), M, N, K, 2, SPLIT_X, None)
        assert rc == 0, lib.mcm_last_error(h)
        torch.cuda.synchronize()
        np.testing.assert_allclose(rd.cpu().numpy(), want + r0, rtol=0, atol=2e-5)
        return
    if epi == 1:
        want = want / (1.0 + np.exp(-1.702 * want))
    # (a) the 16-bit output of the plain epilogues: fp16 rounding of an fp32-accurate value
    y = torch.empty((M, N), device="cuda", dtype=torch.float16)
    rc = lib.mcm_op_linear_ex(h, F16, _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), None, M, N, K, epi, SPLIT_X, None)
    assert rc == 0, lib.mcm_last_error(h)
    # (b) the split output: hi + lo carries the fp32 value
    y2 = torch.zeros((M, 2 * N), device="cuda", dtype=torch.float16)
    rc = lib.mcm_op_linear_ex(h, F16, _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y2), None, M, N, K, epi, SPLIT_X | SPLIT_OUT, None)
    assert rc == 0, lib.mcm_last_error(h)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.float().cpu().numpy(), want, rtol=1.5e-3, atol=1.5e-3)
    got = merge_image(y2.cpu().numpy())
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 if epi == 0 else 4e-5)


def test_split_activation_and_split_weight_together(net):
    """fp32-valued weights: W_hi + W_lo against X_hi + X_lo — four passes per logical K-step through the same K loop."""
    rng = np.random.default_rng(7)
    for M, N, K in [(300, 256, 128), (4096, 768, 768)]:
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        xs, ws = split_image(x), split_image(w)
        want = merge_image(xs) @ merge_image(ws).T
        y2 = torch.zeros((M, 2 * N), device="cuda", dtype=torch.float16)
        xd, wd = _dev(xs), _dev(ws)   # (held: a temporary's memory is reused by the next allocation)
        rc = net._lib.mcm_op_linear_ex(net._h, F16, _ptr(xd), _ptr(wd), None, _ptr(y2), None, M, N, K, 0,
                                       SPLIT_W | SPLIT_X | SPLIT_OUT, None)
        assert rc == 0, net._lib.mcm_last_error(net._h)
        torch.cuda.synchronize()
        np.testing.assert_allclose(merge_image(y2.cpu().numpy()), want, rtol=0, atol=2e-5)
        # and against the fp32 product of the unsplit operands: the split forms lose nothing visible at fp32 round-off
        np.testing.assert_allclose(merge_image(y2.cpu().numpy()), x.astype(np.float64) @ w.astype(np.float64).T, rtol=0, atol=3e-5)


def test_split_flags_are_refused_outside_fp16(net):
    x = torch.zeros((128, 256), device="cuda", dtype=torch.bfloat16)
    w = torch.zeros((128, 128), device="cuda", dtype=torch.bfloat16)
    y = torch.zeros((128, 256), device="cuda", dtype=torch.bfloat16)
    assert net._lib.mcm_op_linear_ex(net._h, 0, _ptr(x), _ptr(w), None, _ptr(y), None, 128, 128, 128, 0, SPLIT_X, None) != 0
    assert net._lib.mcm_op_linear_ex(net._h, 1, _ptr(x), _ptr(w), None, _ptr(y), None, 128, 128, 128, 0, SPLIT_OUT, None) != 0
    assert net._lib.mcm_op_linear_ex(net._h, F16, _ptr(x), _ptr(w), None, _ptr(y), None, 128, 128, 128, 2, SPLIT_OUT, None) != 0


@pytest.mark.parametrize("D", [128, 768, 1024])
def test_layernorm_split_output(net, D):
    from oracle import oracle as orc

    rng = np.random.default_rng(D)
    M = 203
    x = (rng.standard_normal((M, D)) * 2 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    want = orc.layernorm(x, g, b, 1e-5)
    y = torch.zeros((M, 2 * D), device="cuda", dtype=torch.float16)
    xd, gd, bd = _dev(x), _dev(g), _dev(b)
    rc = net._lib.mcm_op_layernorm_split(net._h, _ptr(xd), _ptr(gd), _ptr(bd), _ptr(y), M, D, 1e-5, None)
    assert rc == 0, net._lib.mcm_last_error(net._h)
    torch.cuda.synchronize()
    np.testing.assert_allclose(merge_image(y.cpu().numpy()), want, rtol=1e-5, atol=1e-5)
    # the hi half alone is the fp16 arm's output
    hi = y.cpu().numpy().reshape(M, D // 64, 2, 64)[:, :, 0, :].reshape(M, D).astype(np.float32)
    assert np.abs(hi - want).max() <= 2.0 ** -9


@pytest.mark.parametrize("nseq,L,heads", [(3, 197, 12), (2, 50, 12), (2, 257, 16), (4, 17, 2), (1, 32, 4)])
def test_attention_split(net, nseq, L, heads):
    """Three MFMA passes per product on hi / lo pairs (attention.hip attn_tr_kernel<X2>) against the oracle's fp32 attention:
    fp32 round-off, two orders of magnitude below the fp16 kernel's 3e-3."""
    from oracle import oracle as orc

    rng = np.random.default_rng(L * 31 + heads)
    D = heads * 64
    qkv = rng.standard_normal((nseq * L, 3 * D)).astype(np.float32)
    qkv[:, :2 * D] *= 1.5
    qs = split_image(qkv)
    want = orc.attention(merge_image(qs).astype(np.float32), nseq, L, heads, 64, False)
    out = torch.zeros((nseq * L, 2 * D), device="cuda", dtype=torch.float16)
    qd = _dev(qs)
    rc = net._lib.mcm_op_attention_split(net._h, _ptr(qd), _ptr(out), nseq, L, heads, None)
    assert rc == 0, net._lib.mcm_last_error(net._h)
    torch.cuda.synchronize()
    got = merge_image(out.cpu().numpy())
    np.testing.assert_allclose(got, want, rtol=0, atol=8e-6)  # (the fp32 oracle's own round-off is ~1e-6 at these magnitudes)


def test_attention_split_spiked_logits(net):
    from oracle import oracle as orc

    rng = np.random.default_rng(5)
    nseq, L, heads = 1, 197, 2
    D = heads * 64
    qkv = rng.standard_normal((L, 3 * D)).astype(np.float32)
    qkv[7, :D] *= 20.0
    qkv[100, D:2 * D] *= 10.0
    qs = split_image(qkv)
    want = orc.attention(merge_image(qs).astype(np.float32), nseq, L, heads, 64, False)
    out = torch.zeros((L, 2 * D), device="cuda", dtype=torch.float16)
    qd = _dev(qs)
    assert net._lib.mcm_op_attention_split(net._h, _ptr(qd), _ptr(out), nseq, L, heads, None) == 0
    torch.cuda.synchronize()
    got = merge_image(out.cpu().numpy())
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)


def _towers(ckpt, regime, max_batch):
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.synth import make_token_ids
    from mcm_amd.weights import synth_state_dict

    geo = geometry(ckpt)
    sd = synth_state_dict(geo, 0, regime)
    ids, _ = make_token_ids(100, seed=2)
    n16 = NativeCLIP(geo, sd, precision="fp16", max_batch=max_batch, max_prompt_tokens=100 * ids.shape[1])
    n32 = NativeCLIP(geo, sd, precision="fp32", max_batch=max_batch, max_prompt_tokens=100 * ids.shape[1])
    bank = n32.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
    return geo, n16, n32, bank


@pytest.mark.parametrize("ckpt,regime,B,max_batch", [("B16-2L", "fp16-exact", 21, 64), ("ViT-B/16", "fp16-exact", 96, 64),
                                                     ("ViT-B/16", "fp32", 40, 128), ("ViT-B/32", "fp16-exact", 64, 128),
                                                     ("ViT-L/14", "fp16-exact", 24, 64)])
def test_split_activation_tower_equals_the_fp32_arm_to_fp32_round_off(ckpt, regime, B, max_batch):
    """The whole image tower + scoring tail: the split-activation arm of an fp16 handle against the exact-fp32 arm on the same
    weights and pixels.  fp16 arm: |d score| ~ 1e-8; split-activation arm: the fp32 arm's own round-off (it differs from HF by
    3e-10).  Ragged batches (the pad rows of the 256-row tiles), chunking above mcm_x2_max_batch, uint8 ingest, and the
    fp32-valued regime (split weights AND split activations)."""
    geo, n16, n32, bank = _towers(ckpt, regime, max_batch)
    try:
        assert n16.x2_max_batch == max_batch   # (fp16 handles allocate their activation rows at the split width)
        assert n16.split_weights == (regime == "fp32")
        g = torch.Generator(device="cuda").manual_seed(3)
        px = torch.randn((B, 3, geo.image_size, geo.image_size), device="cuda", generator=g)
        want = n32.score_images(px, bank).double()
        got16 = n16.score_images(px, bank).double()
        got = n16.score_images_x2(px, bank).double()       # B > max_batch in the B/16 case: two chunks
        d16, d2 = float((got16 - want).abs().max()), float((got - want).abs().max())
        print(f"{ckpt} {regime}: |d score| fp16 arm {d16:.2e}, split-activation arm {d2:.2e} (scores ~ {float(want.abs().mean()):.3e})")
        assert d2 <= 2e-9 and d2 <= 0.1 * d16, (d2, d16)
        # features too (the Mahalanobis route and get_image_features consumers)
        f32 = n32.get_image_features(px[:8], normalize=True)
        f2 = n16.get_image_features_x2(px[:8], normalize=True)
        assert float((f2 - f32).abs().max()) <= 2e-6
        # the fp16 arm's results are untouched by an interleaved split call (same workspace)
        again = n16.score_images(px, bank).double()
        assert torch.equal(again, got16)
        # determinism
        assert torch.equal(n16.score_images_x2(px, bank).double(), got)
        # uint8 ingest through the same arm
        u8 = torch.randint(0, 256, (5, geo.image_size, geo.image_size, 3), dtype=torch.uint8, device="cuda", generator=g)
        d8 = float((n16.score_images_x2(u8, bank).double() - n32.score_images(u8, bank).double()).abs().max())
        assert d8 <= 2e-9, d8
        assert n16.saturation_count() == 0
    finally:
        n16.close()
        n32.close()


def test_split_arm_is_refused_by_handles_that_cannot_run_it():
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry("tiny")
    sd = synth_state_dict(geo, 0, "fp16-exact")
    for prec in ("bf16", "fp32"):
        n = NativeCLIP(geo, sd, precision=prec, max_batch=8, max_prompt_tokens=77)
        try:
            assert n.x2_max_batch == 0
            with pytest.raises(RuntimeError):
                n.score_images_x2(torch.zeros((1, 3, geo.image_size, geo.image_size), device="cuda"),
                                  torch.zeros((4, geo.proj_dim), device="cuda"))
        finally:
            n.close()


def test_split_arm_edge_batches():
    """A one-image handle (197 rows padded into one 256-row tile), an empty batch, a one-prompt bank, and more images than
    max_batch (chunked) — each equal to what the fp32 arm gives, to fp32 round-off."""
    geo, n16, n32, bank = _towers("B16-2L", "fp16-exact", 1)
    try:
        assert n16.x2_max_batch == 1
        g = torch.Generator(device="cuda").manual_seed(9)
        px = torch.randn((3, 3, geo.image_size, geo.image_size), device="cuda", generator=g)
        want = n32.score_images(px, bank).double()
        got = n16.score_images_x2(px, bank).double()                     # three chunks of one image
        assert float((got - want).abs().max()) <= 2e-9
        assert n16.score_images_x2(px[:0], bank).shape == (0,)
        one = bank[:1].contiguous()                                      # K = 1: the softmax over one prompt is 1 -> score -1
        s1 = n16.score_images_x2(px, one)
        assert torch.equal(s1, torch.full_like(s1, -1.0))
    finally:
        n16.close()
        n32.close()


def test_x2_workspace_is_sized_by_the_caller():
    """mcm_config.x2_max_batch (ABI 4): 0 = the arm at the full batch, n = at most n images per call (same scores, in
    chunks), < 0 = no split-activation workspace — those calls are refused, the fp16 arm is untouched."""
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.synth import make_pixels, make_token_ids
    from mcm_amd.weights import synth_state_dict

    geo = geometry("B16-2L")
    sd = synth_state_dict(geo, 0, "fp16-exact")
    ids, _ = make_token_ids(7, seed=2)
    px = torch.from_numpy(make_pixels(10, geo.image_size, 7, ood=False, seed=3)[0]).cuda()
    out = {}
    for name, x2 in (("full", None), ("three", 3), ("none", -1)):
        n = NativeCLIP(geo, sd, precision="fp16", max_batch=10, max_prompt_tokens=1024, x2_max_batch=x2)
        try:
            txt = n.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
            assert n.x2_max_batch == {"full": 10, "three": 3, "none": 0}[name]
            out[name] = n.score_images(px, txt).clone()
            if name == "none":
                with pytest.raises(RuntimeError):
                    n.score_images_x2(px, txt)
                rc = n._lib.mcm_score_x2(n._h, px.data_ptr(), 0, 2, txt.data_ptr(), txt.shape[0], 1.0, 0, out[name].data_ptr(), None)
                assert rc != 0 and b"fp16 handle" in n._lib.mcm_last_error(n._h)
            else:
                out[name + "_x2"] = n.score_images_x2(px, txt).clone()
            torch.cuda.synchronize()
            assert n.kernel_faults == 0
        finally:
            n.close()
    assert torch.equal(out["full"], out["three"]) and torch.equal(out["full"], out["none"])      # the fp16 arm: same bits
    assert torch.equal(out["full_x2"], out["three_x2"])                                            # the split arm: chunked = whole
