"""Per-kernel parity on a real MI355X: every HIP kernel of the path, called through the
C ABI's operator-level entry points (include/mcm.h), against the CPU oracle on the same
seeded inputs.  bf16-mode comparisons feed the oracle the bf16-rounded operands, so the
only differences left are fp32 accumulation order and the bf16 rounding of outputs.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


PREC = {"bf16": 0, "fp32": 1, "fp16": 2}
DTYPE = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}
# relative rounding step of a 16-bit output (bf16: 8 significand bits, fp16: 11)
OUT_TOL = {"bf16": 1.2e-2, "fp16": 2e-3}


def _round_to(a: np.ndarray, prec: str) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DTYPE[prec]).float().numpy()


def _bf16_round(a: np.ndarray) -> np.ndarray:
    return _round_to(a, "bf16")


def _tiny(harness):
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.config import geometry
    from mcm_amd.weights import synth_state_dict

    geo = geometry("tiny")
    return NativeCLIP(geo, synth_state_dict(geo, 0), precision="bf16", max_batch=64, max_prompt_tokens=4096,
                      harness=harness)


@pytest.fixture(scope="module")
def tiny_net():
    """A handle of the SHIPPED library (libmcm_hip.so): its own kernel choice, no switches."""
    net = _tiny(False)
    yield net
    net.close()


@pytest.fixture(scope="module")
def harness_net():
    """A handle of libmcm_hip_harness.so (same sources, -DMCM_HARNESS): the A/B kernel arms and the
    mcm_debug_* switches live only there."""
    net = _tiny(True)
    yield net
    net.close()


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("D", [128, 512, 768, 1024])
@pytest.mark.parametrize("prec", ["bf16", "fp32", "fp16"])
def test_layernorm(tiny_net, D, prec):
    from oracle import oracle as orc

    rng = np.random.default_rng(D)
    M = 203
    x = (rng.standard_normal((M, D)) * 2 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    want = orc.layernorm(x, g, b, 1e-5)
    p = PREC[prec]
    y = torch.empty((M, D), device="cuda", dtype=DTYPE[prec])
    xd, gd, bd = _dev(x), _dev(g), _dev(b)
    rc = tiny_net._lib.mcm_op_layernorm(tiny_net._h, p, _ptr(xd), _ptr(gd), _ptr(bd), _ptr(y), M, D,
                                        1e-5, 0, None)
    assert rc == 0, tiny_net._lib.mcm_last_error(tiny_net._h)
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    if prec != "fp32":
        np.testing.assert_allclose(got, want, rtol=OUT_TOL[prec], atol=OUT_TOL[prec])
        ulp = 2 ** -6 if prec == "bf16" else 2 ** -9  # one output ulp at |x| < 4
        assert np.abs(got - _round_to(want, prec)).max() <= ulp
    else:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    # in-place fp32 (pre_layrnorm form)
    z = xd.clone()
    rc = tiny_net._lib.mcm_op_layernorm(tiny_net._h, p, _ptr(z), _ptr(gd), _ptr(bd), _ptr(z), M, D, 1e-5,
                                        1, None)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(z.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


GEMM_SHAPES = [
    (2600, 768, 192),     # > 8 row tiles: every XCD's persistent workgroups get work, 3 K-steps
    (128, 128, 64),       # one tile, one K-step (bf16) / two (fp32)
    (300, 256, 128),      # ragged M
    (591, 2304, 768),     # 3 B/16 images through the QKV projection
    (197, 768, 3072),     # fc2 shape, long K
    (50, 192, 128),       # N not a multiple of the 128 tile (masked columns)
    (1000, 1536, 512),    # text-tower QKV
]


@pytest.fixture(params=[-1, 0, 4, 5, 9, 11], ids=["shipped-policy", "tile128", "persist256x256", "pingpong256x256",
                                                  "pingpong256x256-arms-text", "tile64"])   # (arms 1/2/6/7/8 left the tree in round 6)
def gemm_net(request, tiny_net, harness_net):
    """Every GEMM kernel variant must pass the same parity cases (the shipped policy picks by problem size, so
    small test shapes would otherwise only exercise the tile kernel).  -1 = the shipped library as is; the
    forced variants run in the harness library."""
    if request.param < 0:
        yield tiny_net
        return
    assert harness_net._lib.mcm_debug_gemm_variant(request.param) == 0
    yield harness_net
    harness_net._lib.mcm_debug_gemm_variant(-1)


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("prec", ["bf16", "fp32", "fp16"])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_linear(gemm_net, M, N, K, prec, epi):
    tiny_net = gemm_net
    from oracle import oracle as orc

    rng = np.random.default_rng(M * 7 + N * 3 + K + epi)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)  # asymmetric by construction
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    resid0 = rng.standard_normal((M, N)).astype(np.float32)
    p = PREC[prec]
    if prec != "fp32":
        x, w = _round_to(x, prec), _round_to(w, prec)
    lin = orc.linear(x, w, bias)
    dt = DTYPE[prec]
    xd, wd, bd = _dev(x, dt), _dev(w, dt), _dev(bias)
    y = torch.zeros((M, N), device="cuda", dtype=dt)
    rd = _dev(resid0)
    rc = tiny_net._lib.mcm_op_linear(tiny_net._h, p, _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), _ptr(rd), M, N,
                                     K, epi, None)
    assert rc == 0, tiny_net._lib.mcm_last_error(tiny_net._h)
    torch.cuda.synchronize()
    if epi == 0:
        got, want = y.float().cpu().numpy(), lin
    elif epi == 1:
        got, want = y.float().cpu().numpy(), orc.quick_gelu(lin)
    else:
        got, want = rd.cpu().numpy(), resid0 + lin
    if prec != "fp32" and epi != 2:
        np.testing.assert_allclose(got, want, rtol=OUT_TOL[prec], atol=OUT_TOL[prec])  # 16-bit output rounding
    else:
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)      # fp32 accumulation order


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 768), (768, 512, 3072), (2048, 256, 64),
                                   (300, 256, 128), (1, 512, 256), (700, 768, 768), (129, 256, 3072)])
@pytest.mark.parametrize("prec", ["bf16", "fp32", "fp16"])
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("pp_variant", [5, 9], ids=["shipped-text", "arms-text"])   # (mfma32 / balanced-dma / staggered: removed in round 6)
def test_linear_pingpong_interior_shapes_vs_oracle(harness_net, M, N, K, prec, epi, pp_variant):
    tiny_net = harness_net
    """The ping-pong 256x256 kernel only takes problems made of whole tiles, so it gets its own oracle cases:
    one tile, a few tiles on a few workgroups, a long K, a single K-step.  The shapes with a partial last M tile
    check the routing: under variant 5 they must come out right through the plain persistent kernel.  (Partial
    M tiles inside the ping-pong kernel were built and measured: the second epilogue form cost the whole-tile
    path 3 - 5 %, DESIGN.md section 5.4.)"""
    from oracle import oracle as orc

    rng = np.random.default_rng(M + N * 5 + K * 11 + epi)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    resid0 = rng.standard_normal((M, N)).astype(np.float32)
    if prec != "fp32":
        x, w = _round_to(x, prec), _round_to(w, prec)
    lin = orc.linear(x, w, bias)
    dt = DTYPE[prec]
    xd, wd, bd = _dev(x, dt), _dev(w, dt), _dev(bias)
    y = torch.zeros((M, N), device="cuda", dtype=dt)
    rd = _dev(resid0)
    try:
        assert tiny_net._lib.mcm_debug_gemm_variant(pp_variant) == 0
        rc = tiny_net._lib.mcm_op_linear(tiny_net._h, PREC[prec], _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), _ptr(rd),
                                         M, N, K, epi, None)
        assert rc == 0, tiny_net._lib.mcm_last_error(tiny_net._h)
        torch.cuda.synchronize()
    finally:
        tiny_net._lib.mcm_debug_gemm_variant(-1)
    if epi == 0:
        got, want = y.float().cpu().numpy(), lin
    elif epi == 1:
        got, want = y.float().cpu().numpy(), orc.quick_gelu(lin)
    else:
        got, want = rd.cpu().numpy(), resid0 + lin
    if prec != "fp32" and epi != 2:
        np.testing.assert_allclose(got, want, rtol=OUT_TOL[prec], atol=OUT_TOL[prec])
    else:
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("prec", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("N,K,epi", [(2304, 768, 0), (3072, 768, 1), (768, 3072, 2), (768, 768, 2)])
def test_linear_full_size_variants_bitwise(tiny_net, harness_net, N, K, epi, prec):
    """The four GEMM shapes of a B/16 layer at batch 512 (M = 512*197): the persistent 256x256 kernels
    (both wait forms, and the ping-pong kernel) against the one-workgroup-per-tile kernel, bit for bit, three
    launches each, in every operand format.
    Full-size runs are what exposes timing-dependent faults (missed hazards, DMA/LDS ordering) that
    the small oracle-checked shapes above cannot: the tile kernel shares the fragment and epilogue
    arithmetic but none of the persistent kernel's pipelining."""
    M = 512 * 197 if prec != "fp32" else 25600  # whole 256-row tiles either way (the ping-pong kernel needs them)
    dt = DTYPE[prec]
    g = torch.Generator(device="cuda").manual_seed(N + K + epi)
    x = torch.randn((M, K), generator=g, device="cuda").to(dt)
    w = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(dt)
    bias = 0.1 * torch.randn(N, generator=g, device="cuda")
    resid0 = torch.randn((M, N), generator=g, device="cuda") if epi == 2 else None
    bits = {2: torch.int16, 4: torch.int32}

    def run(variant):  # -1: the shipped library and its own choice; others: forced in the harness library
        net = tiny_net if variant < 0 else harness_net
        if variant >= 0:
            assert net._lib.mcm_debug_gemm_variant(variant) == 0
        y = torch.zeros((M, N), device="cuda", dtype=dt)
        rd = resid0.clone() if epi == 2 else y
        rc = net._lib.mcm_op_linear(net._h, PREC[prec], _ptr(x), _ptr(w), _ptr(bias), _ptr(y),
                                    _ptr(rd), M, N, K, epi, None)
        assert rc == 0, net._lib.mcm_last_error(net._h)
        torch.cuda.synchronize()
        return rd if epi == 2 else y

    try:
        ref = run(0)
        assert torch.isfinite(ref.float()).all()
        for variant in (-1, 3, 4, 5, 9):
            for _ in range(3):
                got = run(variant)
                assert torch.equal(got.view(bits[got.element_size()]), ref.view(bits[ref.element_size()])), \
                    f"variant {variant}"
        # the grouped tile walk (N tiles in groups of g: the W re-fetch A/B of EXPERIMENTS.md) visits the same tiles in
        # another order: same bits, in the arms text of the ping-pong kernel (9) and in the plain persistent kernel (3)
        for gn in (1, 2, 4):
            assert harness_net._lib.mcm_debug_gemm_group_n(gn) == 0
            for variant in (9, 3):
                got = run(variant)
                assert torch.equal(got.view(bits[got.element_size()]), ref.view(bits[ref.element_size()])), \
                    f"group_n {gn} variant {variant}"
    finally:
        harness_net._lib.mcm_debug_gemm_group_n(0)
        harness_net._lib.mcm_debug_gemm_variant(-1)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K,epi", [(25600, 768, 768, 2), (25600, 768, 3072, 2), (25600, 768, 768, 0),
                                       (512 * 257, 1024, 1024, 2)])
def test_linear_sliver_split_bitwise(tiny_net, harness_net, M, N, K, epi, prec):
    """Sliver-round split of launch_gemm (gemm.hip): problems whose tile count is a little more than whole rounds of the
    persistent grid — ViT-B/32 at batch 512 (100 row tiles x 3 = 300 tiles on 256 workgroups) — are cut at a row-tile
    boundary into a ping-pong launch and a tile-kernel launch (not ViT-L/14's: its single left-over row tile makes too
    few workgroups to pay, the last case checks that it still comes out the same).  The shipped library's own choice (split) against
    the forced unsplit ping-pong kernel and the tile kernel of the harness library, bit for bit, incl. the last rows."""
    dt = DTYPE[prec]
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    x = torch.randn((M, K), generator=g, device="cuda").to(dt)
    w = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(dt)
    bias = 0.1 * torch.randn(N, generator=g, device="cuda")
    resid0 = torch.randn((M, N), generator=g, device="cuda") if epi == 2 else None

    def run(variant):
        net = tiny_net if variant < 0 else harness_net
        if variant >= 0:
            assert net._lib.mcm_debug_gemm_variant(variant) == 0
        y = torch.zeros((M, N), device="cuda", dtype=dt)
        rd = resid0.clone() if epi == 2 else y
        rc = net._lib.mcm_op_linear(net._h, PREC[prec], _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(rd), M, N, K, epi, None)
        assert rc == 0, net._lib.mcm_last_error(net._h)
        torch.cuda.synchronize()
        return rd if epi == 2 else y

    try:
        split = run(-1)
        assert torch.isfinite(split.float()).all() and bool((split[-256:].float() != 0).any())
        for variant in (0, 5):
            got = run(variant)
            assert torch.equal(got.view(torch.int32 if epi == 2 else torch.int16),
                               split.view(torch.int32 if epi == 2 else torch.int16)), f"variant {variant}"
    finally:
        harness_net._lib.mcm_debug_gemm_variant(-1)


@pytest.mark.parametrize("prec", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("M,N,K,epi", [(3152, 768, 768, 2), (3152, 2304, 768, 0), (3152, 768, 3072, 2), (197, 768, 768, 2),
                                       (1000, 3072, 768, 1), (77, 512, 512, 0)])
def test_linear_tile64_bitwise(tiny_net, harness_net, M, N, K, epi, prec):
    """The 64x128 tile kernel (shipped since round 4 for problems that give the 128x128 kernel fewer than two workgroups per
    CU; forced everywhere by harness variant 11) against the 128x128 tile kernel (variant 0), bit for bit, ragged row counts
    included — and the shipped library's own choice on these shapes (which is the 64x128 kernel) against both."""
    dt = DTYPE[prec]
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    x = torch.randn((M, K), generator=g, device="cuda").to(dt)
    w = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(dt)
    bias = 0.1 * torch.randn(N, generator=g, device="cuda")
    resid0 = torch.randn((M, N), generator=g, device="cuda") if epi == 2 else None
    net = harness_net

    def run(variant):
        assert net._lib.mcm_debug_gemm_variant(variant) == 0
        y = torch.zeros((M, N), device="cuda", dtype=dt)
        rd = resid0.clone() if epi == 2 else y
        rc = net._lib.mcm_op_linear(net._h, PREC[prec], _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(rd), M, N, K, epi, None)
        assert rc == 0, net._lib.mcm_last_error(net._h)
        torch.cuda.synchronize()
        return rd if epi == 2 else y

    try:
        ref = run(0)
        got = run(11)
        assert torch.isfinite(ref.float()).all()
        view = torch.int32 if ref.element_size() == 4 else torch.int16
        assert torch.equal(got.view(view), ref.view(view))
        y = torch.zeros((M, N), device="cuda", dtype=dt)      # the shipped library, its own size policy
        rd = resid0.clone() if epi == 2 else y
        assert tiny_net._lib.mcm_op_linear(tiny_net._h, PREC[prec], _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(rd), M, N, K,
                                           epi, None) == 0
        torch.cuda.synchronize()
        assert torch.equal((rd if epi == 2 else y).view(view), ref.view(view))
    finally:
        net._lib.mcm_debug_gemm_variant(-1)


@pytest.mark.parametrize("N,K,epi", [(3072, 1024, 0), (4096, 1024, 1), (1024, 4096, 2), (1024, 1024, 2)])
def test_linear_l14_shapes_pingpong_bitwise(tiny_net, harness_net, N, K, epi):
    """BASELINE config 4 (ViT-L/14, batch 256: M = 256 * 257 rows, K = 1024 / 4096): the ping-pong kernel against
    the one-workgroup-per-tile kernel, bit for bit, fp16 operands — other K-loop lengths (16 / 64 steps), other
    tile counts per workgroup and another walk direction than the B/16 cases above."""
    M = 256 * 257
    dt = DTYPE["fp16"]
    g = torch.Generator(device="cuda").manual_seed(N * 3 + K + epi)
    x = torch.randn((M, K), generator=g, device="cuda").to(dt)
    w = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(dt)
    bias = 0.1 * torch.randn(N, generator=g, device="cuda")
    resid0 = torch.randn((M, N), generator=g, device="cuda") if epi == 2 else None

    def run(variant):
        net = tiny_net if variant < 0 else harness_net
        if variant >= 0:
            assert net._lib.mcm_debug_gemm_variant(variant) == 0
        y = torch.zeros((M, N), device="cuda", dtype=dt)
        rd = resid0.clone() if epi == 2 else y
        rc = net._lib.mcm_op_linear(net._h, PREC["fp16"], _ptr(x), _ptr(w), _ptr(bias), _ptr(y),
                                    _ptr(rd), M, N, K, epi, None)
        assert rc == 0, net._lib.mcm_last_error(net._h)
        torch.cuda.synchronize()
        return rd if epi == 2 else y

    try:
        ref = run(0)
        assert torch.isfinite(ref.float()).all()
        for variant in (-1, 5, 9):
            for _ in range(2):  # mcm_op_linear alternates the walk direction per launch: both get exercised
                got = run(variant)
                view = torch.int32 if epi == 2 else torch.int16
                assert torch.equal(got.view(view), ref.view(view)), f"variant {variant}"
    finally:
        harness_net._lib.mcm_debug_gemm_variant(-1)


ATTN_CASES = [(3, 197, 12, False), (2, 50, 12, False), (5, 17, 2, False), (4, 77, 8, True),
              (6, 16, 8, True), (2, 257, 16, False), (3, 33, 2, True)]


@pytest.mark.parametrize("nseq,L,heads,causal", ATTN_CASES)
@pytest.mark.parametrize("prec", ["bf16", "fp32", "fp16"])
def test_attention(tiny_net, nseq, L, heads, causal, prec):
    from oracle import oracle as orc

    rng = np.random.default_rng(L * 31 + heads)
    D = heads * 64
    qkv = rng.standard_normal((nseq * L, 3 * D)).astype(np.float32)
    qkv[:, :2 * D] *= 1.5  # O(1)-spread logits after the 0.125 scale: a non-uniform softmax
    p = PREC[prec]
    if prec != "fp32":
        qkv = _round_to(qkv, prec)
    want = orc.attention(qkv, nseq, L, heads, 64, causal)
    dt = DTYPE[prec]
    qd = _dev(qkv, dt)
    out = torch.zeros((nseq * L, D), device="cuda", dtype=dt)
    rc = tiny_net._lib.mcm_op_attention(tiny_net._h, p, _ptr(qd), _ptr(out), nseq, L, heads, int(causal), None)
    assert rc == 0, tiny_net._lib.mcm_last_error(tiny_net._h)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    if prec == "bf16":
        np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2)  # bf16 P and bf16 output
    elif prec == "fp16":
        np.testing.assert_allclose(got, want, rtol=3e-3, atol=3e-3)  # fp16 P and fp16 output
    else:
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def test_attention_spiked_logits(tiny_net):
    """One key dominating a row (softmax ~ one-hot) and large negative logits elsewhere."""
    from oracle import oracle as orc

    rng = np.random.default_rng(5)
    nseq, L, heads = 1, 197, 2
    D = heads * 64
    qkv = rng.standard_normal((L, 3 * D)).astype(np.float32)
    qkv[7, :D] *= 20.0          # query 7 is huge
    qkv[100, D:2 * D] *= 10.0   # key 100 is huge
    qkv = _bf16_round(qkv)
    want = orc.attention(qkv, nseq, L, heads, 64, False)
    qd = _dev(qkv, torch.bfloat16)
    out = torch.zeros((L, D), device="cuda", dtype=torch.bfloat16)
    rc = tiny_net._lib.mcm_op_attention(tiny_net._h, 0, _ptr(qd), _ptr(out), nseq, L, heads, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("K", [10, 37, 100, 1000])
@pytest.mark.parametrize("T", [1.0, 2.0, 0.01])
def test_score_kinds(tiny_net, K, T):
    from mcm_amd.config import SCORE_KINDS
    from oracle import oracle as orc

    rng = np.random.default_rng(K)
    P, B = tiny_net.geo.proj_dim, 33
    img = rng.standard_normal((B, P)).astype(np.float32)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    txt = rng.standard_normal((K, P)).astype(np.float32)
    txt /= np.linalg.norm(txt, axis=1, keepdims=True)
    for name, kind in SCORE_KINDS.items():
        want = orc.score_features(img, txt, T, kind)
        got = tiny_net.score_features(_dev(img), _dev(txt), T, name).cpu().numpy()
        tol = dict(rtol=3e-5, atol=1e-6) if name != "var" else dict(rtol=2e-3, atol=1e-10)
        np.testing.assert_allclose(got, want, err_msg=f"{name} K={K} T={T}", **tol)


ATTN_MORE = [(2, 65, 2, False), (2, 80, 2, False), (2, 96, 2, False), (2, 112, 2, False), (3, 40, 2, True),
             (2, 128, 2, True), (1, 288, 1, False), (2, 1, 2, False), (8, 65, 2, False), (16, 40, 3, True)]


@pytest.mark.parametrize("nseq,L,heads,causal", ATTN_MORE)
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_attention_every_tile_count(tiny_net, harness_net, nseq, L, heads, causal, prec):
    """The transpose-read kernel is instantiated per key-tile count (1 .. 18 tiles of 16 keys); odd counts
    end with a half-empty key step.  Both 16-bit modes, the round-2 kernel AND the round-1 kernel (A/B arm),
    against the oracle."""
    from oracle import oracle as orc

    rng = np.random.default_rng(L * 7 + heads)
    D = heads * 64
    qkv = rng.standard_normal((nseq * L, 3 * D)).astype(np.float32)
    qkv[:, :2 * D] *= 1.5
    qkv = _round_to(qkv, prec)
    want = orc.attention(qkv, nseq, L, heads, 64, causal)
    dt = DTYPE[prec]
    qd = _dev(qkv, dt)
    tol = 2e-2 if prec == "bf16" else 3e-3
    outs = {}
    try:
        # -1: the shipped library; 1 / 0: both kernels forced in the harness library; 10: the XCD-aware deal of the
        # workgroups (sequence counts that are multiples of 8; the plain deal otherwise)
        for variant in (-1, 1, 0, 10, 11):  # 11: the rotated deal of the q-blocks to the waves (SIMD balance)
            net = tiny_net if variant < 0 else harness_net
            if variant >= 0:
                assert net._lib.mcm_debug_attention_variant(variant) == 0
            out = torch.zeros((nseq * L, D), device="cuda", dtype=dt)
            rc = net._lib.mcm_op_attention(net._h, PREC[prec], _ptr(qd), _ptr(out), nseq, L, heads, int(causal), None)
            assert rc == 0, net._lib.mcm_last_error(net._h)
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.float().cpu().numpy(), want, rtol=tol, atol=tol,
                                       err_msg=f"variant {variant}")
            outs[variant] = out
        assert torch.equal(outs[1], outs[10]) and torch.equal(outs[1], outs[-1]) and torch.equal(outs[1], outs[11])
    finally:
        harness_net._lib.mcm_debug_attention_variant(1)


PS_CASES = [(1, 197, 12, None), (7, 197, 12, None), (43, 197, 12, None), (30, 193, 12, None), (25, 208, 3, None),
            (9, 200, 16, None), (40, 197, 12, 1), (33, 197, 12, 40), (400, 197, 12, None)]


@pytest.mark.parametrize("nseq,L,heads,qrows", PS_CASES)
@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_attention_persistent_form_bitwise(harness_net, nseq, L, heads, qrows, prec):
    """The persistent form of the 13-tile kernel (loader waves + compute waves, one workgroup per CU: DESIGN.md 4.2) forced at
    every size — fewer jobs than CUs, job counts that are no multiple of the grid, every valid key count of the 13-tile class,
    reduced query rows (mcm_op_attention_q), both walk directions — against the 8-wave kernel, bit for bit, and against the
    oracle for the small cases.  (Both kernels run a q-block's arithmetic in the same order.)"""
    from oracle import oracle as orc

    lib = harness_net._lib
    D = heads * 64
    dt = DTYPE[prec]
    g = torch.Generator(device="cuda").manual_seed(L * 1000 + nseq)
    qkv = (torch.randn((nseq * L, 3 * D), device="cuda", generator=g) * 1.4).to(dt)
    outs = {}
    try:
        for variant in (36, 21):
            assert lib.mcm_debug_attention_variant(variant) == 0
            for rep in range(2):   # both walk directions
                out = torch.zeros((nseq * L, D), device="cuda", dtype=dt)
                rc = lib.mcm_debug_op_attention(harness_net._h, PREC[prec], _ptr(qkv), _ptr(out), nseq, L, heads, 0, qrows or 0,
                                                rep, None)
                assert rc == 0, lib.mcm_last_error(harness_net._h)
                torch.cuda.synchronize()
                outs[(variant, rep)] = out
        assert torch.equal(outs[(36, 0)], outs[(36, 1)])
        for rep in range(2):
            assert torch.equal(outs[(21, rep)].view(torch.int16), outs[(36, 0)].view(torch.int16)), f"walk {rep}"
    finally:
        lib.mcm_debug_attention_variant(1)
    if nseq <= 9 and qrows is None:
        want = orc.attention(qkv.float().cpu().numpy(), nseq, L, heads, 64, False)
        tol = 2e-2 if prec == "bf16" else 3e-3
        np.testing.assert_allclose(outs[(21, 0)].float().cpu().numpy(), want, rtol=tol, atol=tol)


def test_attention_persistent_form_waits_are_bounded(harness_net):
    """every wait of the persistent attention kernel is bounded.  With the poll budget forced to 0 (harness
    switch) the compute waves' very first wait for a job gives up: the launch ENDS (no hang), the handle's sticky fault word is
    set, every later compute call returns MCM_EHIP and says why; with the word cleared and the shipped budget restored the same
    launch is bit-identical to the 8-wave kernel again and reports no fault."""
    lib, h = harness_net._lib, harness_net._h
    nseq, L, heads = 43, 197, 12
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = (torch.randn((nseq * L, 3 * D), device="cuda", generator=g) * 1.4).to(torch.float16)

    def run(variant):
        assert lib.mcm_debug_attention_variant(variant) == 0
        out = torch.zeros((nseq * L, D), device="cuda", dtype=torch.float16)
        rc = lib.mcm_debug_op_attention(h, PREC["fp16"], _ptr(qkv), _ptr(out), nseq, L, heads, 0, 0, 0, None)
        assert rc == 0, lib.mcm_last_error(h)
        torch.cuda.synchronize()
        return out

    try:
        assert lib.mcm_kernel_faults(h) == 0
        want = run(36)
        assert lib.mcm_debug_attn_spin_budget(0) == 0
        run(21)                                  # returns: the kernel gave up instead of waiting
        assert lib.mcm_kernel_faults(h) != 0
        px = torch.zeros((1, 3, harness_net.geo.image_size, harness_net.geo.image_size), device="cuda")
        with pytest.raises(RuntimeError, match="gave up a wait"):
            harness_net.get_image_features(pixel_values=px)
        assert lib.mcm_debug_attn_spin_budget(1 << 22) == 0
        assert lib.mcm_debug_clear_faults(h) == 0 and lib.mcm_kernel_faults(h) == 0
        got = run(21)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)) and lib.mcm_kernel_faults(h) == 0
        harness_net.get_image_features(pixel_values=px)   # the handle works again
    finally:
        lib.mcm_debug_attn_spin_budget(1 << 22)
        lib.mcm_debug_clear_faults(h)
        lib.mcm_debug_attention_variant(1)


def test_attention_full_size_bitwise_repeatable(tiny_net):
    """B/16 batch 512 (6144 workgroups, 3 per CU): three launches bit-identical, and equal to the same
    sequences run as a small launch (a timing-dependent fault shows up as a differing workgroup)."""
    nseq, L, heads = 512, 197, 12
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = (torch.randn((nseq * L, 3 * D), device="cuda", generator=g) * 1.3).to(torch.float16)
    outs = []
    for _ in range(3):
        out = torch.zeros((nseq * L, D), device="cuda", dtype=torch.float16)
        rc = tiny_net._lib.mcm_op_attention(tiny_net._h, 2, _ptr(qkv), _ptr(out), nseq, L, heads, 0, None)
        assert rc == 0
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    small = torch.zeros((7 * L, D), device="cuda", dtype=torch.float16)
    rc = tiny_net._lib.mcm_op_attention(tiny_net._h, 2, _ptr(qkv[100 * L:107 * L].contiguous()), _ptr(small), 7, L,
                                        heads, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(small, outs[0][100 * L:107 * L])
    assert tiny_net.kernel_faults == 0   # the persistent form's bounded waits never ran out (include/mcm.h mcm_kernel_faults)


@pytest.mark.parametrize("epi", [0, 1])
def test_fp16_outputs_saturate_instead_of_overflowing(tiny_net, epi):
    """MODE.FP16_OVFL: a QKV / QuickGELU output beyond the fp16 range is stored as +-65504, not +-inf (an inf
    would turn the whole row into NaN at the next LayerNorm).  In-range outputs are untouched."""
    from oracle import oracle as orc

    M, N, K = 70, 128, 64
    rng = np.random.default_rng(1)
    x = _round_to(rng.standard_normal((M, K)), "fp16")
    w = _round_to(rng.standard_normal((N, K)) * 0.1, "fp16")
    b = np.zeros(N, np.float32)
    x[3, :] = 200.0
    w[5, :] = 60.0      # row 3 x col 5: 200*60*64 = 768000 > 65504
    w[6, :] = -60.0     # and -768000 (QuickGELU maps it to ~0)
    want = orc.linear(x, w, b)
    if epi == 1:
        want = want / (1.0 + np.exp(-1.702 * want.astype(np.float64)))
    y = torch.zeros((M, N), device="cuda", dtype=torch.float16)
    xd, wd, bd = _dev(x, torch.float16), _dev(w, torch.float16), _dev(b)   # keep the operands alive
    tiny_net.saturation_count(reset=True)
    rc = tiny_net._lib.mcm_op_linear(tiny_net._h, 2, _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), None, M, N, K, epi, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = y.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert got[3, 5] == 65504.0
    if epi == 0:
        assert got[3, 6] == -65504.0
    ok = np.abs(want) < 6e4
    np.testing.assert_allclose(got[ok], np.asarray(want, np.float32)[ok], rtol=2e-3, atol=2e-3)
    # ... and not silently: the handle's sticky counter saw it (mcm_saturation_count), once, then reads 0 again
    assert tiny_net.saturation_count(reset=True) >= 1
    assert tiny_net.saturation_count() == 0
    with pytest.warns(RuntimeWarning, match="saturated"):
        tiny_net._lib.mcm_op_linear(tiny_net._h, 2, _ptr(xd), _ptr(wd), _ptr(bd), _ptr(y), None, M, N, K, epi, None)
        assert tiny_net.warn_if_saturated("a test") >= 1
    # the same product in bf16 mode (fp32 exponent range) and an in-range fp16 launch report nothing
    yb = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
    xb, wb = _dev(x, torch.bfloat16), _dev(w, torch.bfloat16)
    assert tiny_net._lib.mcm_op_linear(tiny_net._h, 0, _ptr(xb), _ptr(wb), _ptr(bd), _ptr(yb), None, M, N, K, epi, None) == 0
    xs = _dev(x * 1e-3, torch.float16)
    assert tiny_net._lib.mcm_op_linear(tiny_net._h, 2, _ptr(xs), _ptr(wd), _ptr(bd), _ptr(y), None, M, N, K, epi, None) == 0
    assert tiny_net.saturation_count() == 0


def test_fp16_saturation_counter_in_the_persistent_gemm_kernels(tiny_net, harness_net):
    """The same watch in the 256x256 persistent kernels (plain and ping-pong), on a problem made of whole tiles with
    one saturating element, through the shipped library (its own kernel choice) and every forced variant."""
    M, N, K = 1024, 512, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.float16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.float16)
    bias = torch.zeros(N, device="cuda")
    y = torch.zeros((M, N), device="cuda", dtype=torch.float16)
    xs = x.clone()
    xs[777, :] = 250.0
    ws = w.clone()
    ws[300, :] = 250.0     # 250 * 250 * 128 = 8e6
    try:
        for variant in (-1, 0, 3, 5, 9):
            net = tiny_net if variant < 0 else harness_net
            if variant >= 0:
                assert net._lib.mcm_debug_gemm_variant(variant) == 0
            net.saturation_count(reset=True)
            assert net._lib.mcm_op_linear(net._h, 2, _ptr(x), _ptr(w), _ptr(bias), _ptr(y), None, M, N, K, 0, None) == 0
            assert net.saturation_count() == 0, variant
            assert net._lib.mcm_op_linear(net._h, 2, _ptr(xs), _ptr(ws), _ptr(bias), _ptr(y), None, M, N, K, 0, None) == 0
            assert net.saturation_count() >= 1, variant
            torch.cuda.synchronize()
            assert float(y[777, 300]) == 65504.0 and torch.isfinite(y).all()
    finally:
        harness_net._lib.mcm_debug_gemm_variant(-1)


def test_fp16_layernorm_and_attention_saturate(tiny_net):
    lib = tiny_net._lib
    x = torch.zeros((8, 128), device="cuda")
    x[:, 0] = 1.0
    gamma = torch.full((128,), 3e4, device="cuda")   # (x - mean) / std ~ 11.3 at column 0 -> 3.4e5
    beta = torch.zeros(128, device="cuda")
    y = torch.zeros((8, 128), device="cuda", dtype=torch.float16)
    tiny_net.saturation_count(reset=True)
    assert lib.mcm_op_layernorm(tiny_net._h, 2, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), 8, 128, 1e-5, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and float(y[0, 0]) == 65504.0
    assert tiny_net.saturation_count() >= 1   # the LayerNorm reports it too


@pytest.mark.parametrize("P,geo_name", [(512, "ViT-B/16"), (768, "ViT-L/14")])
@pytest.mark.parametrize("K", [1000, 37])
def test_score_kinds_full_projection_dims(P, geo_name, K):
    """The scoring tail at the projection widths of the real checkpoints (512 for B/16 and B/32, 768 for
    L/14) and K = 1000, all five --score kinds, against the oracle (test_score_kinds covers the tiny P)."""
    import dataclasses

    from mcm_amd.config import SCORE_KINDS, geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict
    from oracle import oracle as orc

    geo = dataclasses.replace(geometry("tiny"), name=f"tiny-P{P}", proj_dim=P)
    net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="fp16", max_batch=8, max_prompt_tokens=256)
    try:
        rng = np.random.default_rng(K + P)
        B = 41
        img = rng.standard_normal((B, P)).astype(np.float32)
        img /= np.linalg.norm(img, axis=1, keepdims=True)
        txt = rng.standard_normal((K, P)).astype(np.float32)
        txt /= np.linalg.norm(txt, axis=1, keepdims=True)
        for T in (1.0, 0.05):
            for name, kind in SCORE_KINDS.items():
                want = orc.score_features(img, txt, T, kind)
                got = net.score_features(_dev(img), _dev(txt), T, name).cpu().numpy()
                tol = dict(rtol=3e-5, atol=1e-6) if name != "var" else dict(rtol=2e-3, atol=1e-10)
                np.testing.assert_allclose(got, want, err_msg=f"{name} K={K} T={T} P={P}", **tol)
        with pytest.raises(ValueError):   # a bank of the wrong width is refused, not read out of bounds
            net.score_features(_dev(img), _dev(txt[:, : P // 2].copy()), 1.0, "MCM")
        with pytest.raises(ValueError):
            net.score_images(torch.zeros((1, 3, geo.image_size, geo.image_size), device="cuda"),
                             _dev(txt[:, : P // 2].copy()))
    finally:
        net.close()


def test_device_histogram_matches_numpy(tiny_net):
    rng = np.random.default_rng(9)
    s = rng.standard_normal(50001).astype(np.float32)
    edges = np.linspace(-3, 3, 257).astype(np.float32)
    s[:5] = [edges[0], edges[-1], edges[7], -3.5, 9.0]      # edge values and out-of-range scores
    got = tiny_net.histogram(_dev(s), edges).cpu().numpy()
    want = np.histogram(s, bins=edges)[0]
    assert got.dtype == np.int64 and np.array_equal(got, want)
