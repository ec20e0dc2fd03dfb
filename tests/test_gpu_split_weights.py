"""Split-weight arm (include/mcm.h MCM_WEIGHTS_*; gemm.hip "Split weights") on a real MI355X.

A 16-bit vision mode holds a GEMM weight either as ONE operand — lossless when the weight is an fp16 (bf16) number,
the case of the reference's checkpoints — or as W_hi + W_lo, two operands met by the same staged X K-step, which is
exact for the weight operand whatever its values.  Checked here: the GEMM kernels in split form against float64 on
fp32-valued weights (every kernel the size policy can pick), the policy (`auto` splits exactly when a weight is not
a number of the operand dtype), the dtype argument of mcm_set_weight, and the towers in split mode against the HF
fixtures and the exact-fp32 arm.
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

PREC = {"bf16": 0, "fp32": 1, "fp16": 2}
DTYPE = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def net():
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry("tiny")
    n = NativeCLIP(geo, synth_state_dict(geo, 0), precision="fp16", max_batch=64, max_prompt_tokens=4096)
    yield n
    n.close()


# (M, N, K): tile kernel (small), plain persistent kernel (ragged, > half a round), ping-pong (whole tiles, > 128 of
# them), the sliver split (a little more than whole rounds), a single K-step, a long K
SHAPES = [(300, 256, 128), (50, 192, 64), (197, 768, 3072), (2600 * 8, 768, 192), (256 * 140, 256, 768),
          (256 * 65, 768, 768), (25600, 768, 768), (4096, 2304, 768)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("epi", [0, 2])
def test_split_linear_is_exact_for_the_weight_operand(net, M, N, K, prec, epi):
    """y = x W^T with fp32-valued W (nothing like an fp16 number) and 16-bit x: the split form must agree with the
    float64 product of the SAME x and the FULL-precision W to fp32-accumulation accuracy, where the single-operand
    form is off by the weight rounding (2^-12 per element for fp16, 2^-9 for bf16)."""
    dt = DTYPE[prec]
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K + epi)
    x = torch.randn((M, K), generator=g, device="cuda").to(dt)
    w = torch.randn((N, K), generator=g, device="cuda") * 0.02          # fp32-valued, |w| ~ 2^-6: lo is fp16-subnormal
    bias = 0.1 * torch.randn(N, generator=g, device="cuda")
    resid0 = torch.randn((M, N), generator=g, device="cuda")
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(1))[:64].cuda()  # float64 reference on 64 rows
    want = x[rows].double() @ w.double().T + bias.double()
    if epi == 2:
        want = want + resid0[rows].double()
    scale = float(want.abs().max())

    def run(split):
        y = torch.zeros((M, N), device="cuda", dtype=torch.float32 if epi == 2 else dt)
        rd = resid0.clone() if epi == 2 else None
        if split:
            wd = torch.empty((N, 2 * K), device="cuda", dtype=dt)
            assert net._lib.mcm_op_split_weight(net._h, PREC[prec], _ptr(w), N, K, _ptr(wd), None) == 0
        else:
            wd = w.to(dt)
        rc = net._lib.mcm_op_linear_ex(net._h, PREC[prec], _ptr(x), _ptr(wd), _ptr(bias), _ptr(y), _ptr(rd), M, N, K,
                                       epi, 1 if split else 0, None)
        assert rc == 0, net._lib.mcm_last_error(net._h)
        torch.cuda.synchronize()
        return (rd if epi == 2 else y)[rows].double()

    e_split = float((run(True) - want).abs().max()) / scale
    e_single = float((run(False) - want).abs().max()) / scale
    if epi == 2:  # fp32 output: the split form is at accumulation accuracy, the single form at weight-rounding accuracy
        # hi + lo carries 22 (fp16: lo subnormal -> 2^-24 absolute) / 16 (bf16) significand bits of w
        assert e_split < (3e-6 if prec == "fp16" else 4e-5), (e_split, e_single)
        assert e_single > 8 * e_split, (e_split, e_single)
    else:         # 16-bit output: both are dominated by the output rounding; the split form must not be worse
        ulp = 2.0 ** -11 if prec == "fp16" else 2.0 ** -8
        assert e_split <= ulp * 1.01, e_split
        assert e_split <= e_single * 1.05 + 1e-9


def test_split_image_layout_and_values(net):
    """mcm_op_split_weight: per 64-column K-step hi[64] then lo[64]; hi = fp16(w), lo = fp16(w - hi); hi + lo
    reproduces w to 2^-24 absolute (lo is subnormal for |w| < 2^-2)."""
    N, K = 48, 192
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.randn((N, K), generator=g, device="cuda") * 0.05
    out = torch.empty((N, 2 * K), device="cuda", dtype=torch.float16)
    assert net._lib.mcm_op_split_weight(net._h, PREC["fp16"], _ptr(w), N, K, _ptr(out), None) == 0
    torch.cuda.synchronize()
    o = out.view(N, K // 64, 2, 64)
    hi, lo = o[:, :, 0, :].reshape(N, K), o[:, :, 1, :].reshape(N, K)
    assert torch.equal(hi, w.to(torch.float16))
    assert torch.equal(lo, (w - hi.float()).to(torch.float16))
    assert float((hi.double() + lo.double() - w.double()).abs().max()) <= 2.0 ** -25 * 1.01


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_auto_policy_splits_exactly_when_a_weight_is_not_an_operand_number(precision):
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry("tiny")
    sd = synth_state_dict(geo, 0)
    npdt = np.float16
    rounded = {k: (torch.from_numpy(v).to(DTYPE[precision]).float().numpy()) for k, v in sd.items()}
    n_gemm = sum(v.size for k, v in sd.items() if k.startswith("vision_model.") and v.ndim >= 2 and
                 ("_proj.weight" in k or ".mlp.fc" in k or "patch_embedding.weight" in k))
    a = NativeCLIP(geo, sd, precision=precision, max_batch=8, max_prompt_tokens=1024)
    b = NativeCLIP(geo, rounded, precision=precision, max_batch=8, max_prompt_tokens=1024)
    c = NativeCLIP(geo, sd, precision=precision, max_batch=8, max_prompt_tokens=1024, weight_operands="single")
    d = NativeCLIP(geo, rounded, precision=precision, max_batch=8, max_prompt_tokens=1024, weight_operands="split")
    e = NativeCLIP(geo, sd, precision="fp32", max_batch=8, max_prompt_tokens=1024)
    try:
        assert a.split_weights and 0.9 * n_gemm < a.weights_inexact <= n_gemm
        assert not b.split_weights and b.weights_inexact == 0
        assert not c.split_weights and c.weights_inexact == a.weights_inexact
        assert d.split_weights and d.weights_inexact == 0
        assert not e.split_weights and e.weights_inexact == 0
        if precision == "fp16":  # an fp16 checkpoint handed over AS fp16 (mcm_set_weight dtype = MCM_DT_F16): same handle
            f = NativeCLIP(geo, {k: v.astype(npdt) for k, v in rounded.items()}, precision="fp16", max_batch=8,
                           max_prompt_tokens=1024)
            try:
                assert not f.split_weights and f.weights_inexact == 0
                px = torch.randn((5, 3, geo.image_size, geo.image_size), device="cuda",
                                 generator=torch.Generator(device="cuda").manual_seed(3))
                assert torch.equal(f.get_image_features(px), b.get_image_features(px))
            finally:
                f.close()
        # split mode on weights that ARE operand numbers: lo = 0, the same products in the same order -> same bits
        px = torch.randn((5, 3, geo.image_size, geo.image_size), device="cuda",
                         generator=torch.Generator(device="cuda").manual_seed(4))
        assert torch.equal(d.get_image_features(px), b.get_image_features(px))
    finally:
        for n in (a, b, c, d, e):
            n.close()


@pytest.mark.parametrize("name", ["tiny", "B16-2L"])
def test_split_towers_close_the_weight_rounding_gap(name):
    """fp32-valued seeded weights: image features of the split fp16 arm against the exact-fp32 arm must be as close as
    the single-operand fp16 arm is on fp16-EXACT weights (activation rounding only), and clearly closer than the
    single-operand arm on the same fp32-valued weights."""
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry(name)
    sd = synth_state_dict(geo, 0)
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    B = 24
    px = torch.randn((B, 3, geo.image_size, geo.image_size), device="cuda",
                     generator=torch.Generator(device="cuda").manual_seed(11))

    def feats(weights, precision, **kw):
        n = NativeCLIP(geo, weights, precision=precision, max_batch=B, max_prompt_tokens=1024, **kw)
        try:
            return n.get_image_features(px, normalize=True).double(), n.split_weights
        finally:
            n.close()

    ref, _ = feats(sd, "fp32")
    ref16, _ = feats(sd16, "fp32")
    split, was_split = feats(sd, "fp16")                       # auto -> split
    single, _ = feats(sd, "fp16", weight_operands="single")
    exact16, was16 = feats(sd16, "fp16")                       # auto -> single (weights are fp16 numbers)
    assert was_split and not was16
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt())  # noqa: E731
    e_split, e_single, e_floor = rms(split, ref), rms(single, ref), rms(exact16, ref16)
    assert e_split < 1.5 * e_floor, (e_split, e_single, e_floor)
    assert e_split < 0.8 * e_single, (e_split, e_single, e_floor)


def test_split_mode_scores_vs_oracle():
    """The whole hot path (both towers, MCM tail) in split fp16 mode against the CPU oracle on fp32-valued weights."""
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.synth import make_pixels, make_token_ids
    from mcm_amd.weights import synth_state_dict
    from oracle import oracle as orc

    geo = geometry("tiny")
    sd = synth_state_dict(geo, 0)
    K, B = 10, 21  # ragged batch: 21 * 17 rows
    ids, mask = make_token_ids(K, seed=2)
    px, _ = make_pixels(B, geo.image_size, K, ood=False, seed=1)
    o = orc.OracleCLIP(geo, sd)
    want = orc.score_features(o.encode_image(px), o.encode_text(ids), 1.0, 0)
    net = NativeCLIP(geo, sd, precision="fp16", max_batch=32, max_prompt_tokens=1024)
    try:
        assert net.split_weights
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        got = net.score_images(torch.from_numpy(px).cuda(), txt, 1.0, "MCM").cpu().numpy()
        assert net.saturation_count() == 0
    finally:
        net.close()
    assert np.abs(got - want).max() < 1e-4


def test_text_then_ragged_vision_batch_does_not_trip_the_saturation_watch():
    """the activation buffers are shared by the fp32 text tower and the 16-bit vision tower, and gemm() runs
    ragged batches on rows padded to whole 256-row tiles: the pad rows used to hold whatever the text tower left there
    (fp32 bit patterns read as fp16: inf / NaN) and bumped the sticky saturation counter."""
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.synth import make_token_ids
    from mcm_amd.weights import synth_state_dict

    geo = geometry("B16-2L")
    sd = {k: v.astype(np.float16).astype(np.float32) for k, v in synth_state_dict(geo, 0).items()}
    K = 300
    ids, _ = make_token_ids(K, seed=2)
    net = NativeCLIP(geo, sd, precision="fp16", max_batch=100, max_prompt_tokens=K * ids.shape[1])
    try:
        for B in (100, 37, 3):  # 19700, 7289, 591 rows: none a multiple of 256
            txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)  # dirties the shared buffers
            px = torch.randn((B, 3, 224, 224), device="cuda", generator=torch.Generator(device="cuda").manual_seed(B))
            s1 = net.score_images(px, txt, 1.0, "MCM")
            assert torch.isfinite(s1).all()
            assert net.saturation_count() == 0, B
            s2 = net.score_images(px, txt, 1.0, "MCM")
            assert torch.equal(s1, s2)
    finally:
        net.close()
