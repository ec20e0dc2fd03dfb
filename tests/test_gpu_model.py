"""End-to-end parity of the HIP path on a real MI355X, through the C ABI:

  * towers vs the oracle and vs the committed HF fixtures (tests/golden/clip_*.npz);
  * `get_ood_scores_clip` vs the reference's own outputs (tests/golden/scores_tiny.npz) for
    all five --score kinds, and AUROC / AUPR / FPR95 vs the reference's get_measures;
  * size-independent properties at BASELINE.json's full sizes (B/16, batch 512, K=1000):
    batch-split invariance, prompt-permutation invariance, determinism, shard-and-gather.

Tolerances: fp32 mode (exact-fp32 MFMA, the parity arm) must agree with the fp32 reference
to fp32 round-off; bf16 mode (the benchmarked mode) is held to bf16 operand round-off on
features and to the north-star bar |ΔAUROC|,|ΔFPR95| ≤ 1e-4 where stated.
"""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from mcm_amd.config import SCORE_KINDS, geometry  # noqa: E402
from mcm_amd.synth import (SyntheticImageSet, SyntheticLoader, class_names, make_pixels,  # noqa: E402
                           make_token_ids)
from mcm_amd.weights import synth_state_dict  # noqa: E402


def _net(name, precision, regime="fp32", **kw):
    """regime "fp32": the seeded weights as drawn (a 16-bit arm runs the split-weight GEMMs); "fp16-exact": rounded to
    fp16 values like the reference's checkpoints (one operand per weight)."""
    from mcm_amd.engine import NativeCLIP

    geo = geometry(name)
    return NativeCLIP(geo, synth_state_dict(geo, 0, regime), precision=precision, **kw)


def _unit(a):
    return a / np.linalg.norm(a, axis=-1, keepdims=True)


def _cos(a, b):
    return np.sum(_unit(a) * _unit(b), axis=-1)


@pytest.mark.parametrize("name,fixture,n_extra", [("tiny", "clip_tiny.npz", 29),
                                                  ("B16-2L", "clip_B16-2L.npz", 3),
                                                  ("ViT-B/16", "clip_ViT-B_16.npz", 0),
                                                  ("ViT-B/32", "clip_ViT-B_32.npz", 0),
                                                  ("ViT-L/14", "clip_ViT-L_14.npz", 0)])
@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_towers_vs_hf_fixture(golden_dir, name, fixture, n_extra, precision):
    g = np.load(os.path.join(golden_dir, fixture))
    geo = geometry(name)
    net = _net(name, precision, max_batch=32, max_prompt_tokens=2048)
    try:
        px, _ = make_pixels(int(g["n_img"]), geo.image_size, 10, ood=False, seed=1)
        ids, mask = make_token_ids(int(g["n_txt"]), seed=2)
        pxd, idt, mkt = torch.from_numpy(px).cuda(), torch.from_numpy(ids), torch.from_numpy(mask)
        # the plain contract returns what HF returns: the projection output, not unit-norm
        raw_i = net.get_image_features(pixel_values=pxd).cpu().numpy()
        raw_t = net.get_text_features(input_ids=idt, attention_mask=mkt).cpu().numpy()
        img = net.get_image_features(pixel_values=pxd, normalize=True).cpu().numpy()
        txt = net.get_text_features(input_ids=idt, attention_mask=mkt, normalize=True).cpu().numpy()
        want_i, want_t = _unit(g["image_features"]), _unit(g["text_features"])
        np.testing.assert_allclose(np.linalg.norm(img, axis=1), 1.0, atol=1e-5)
        np.testing.assert_allclose(img, _unit(raw_i), rtol=0, atol=1e-6)   # fused `/= norm` == the reference's
        np.testing.assert_allclose(txt, _unit(raw_t), rtol=0, atol=1e-6)
        if precision == "fp32":
            sc_i, sc_t = np.abs(g["image_features"]).max(), np.abs(g["text_features"]).max()
            np.testing.assert_allclose(raw_i, g["image_features"], rtol=0, atol=2e-4 * max(1.0, sc_i))
            np.testing.assert_allclose(raw_t, g["text_features"], rtol=0, atol=2e-4 * max(1.0, sc_t))
            np.testing.assert_allclose(img, want_i, rtol=0, atol=2e-4)
            np.testing.assert_allclose(txt, want_t, rtol=0, atol=2e-4)
        else:
            floor = 0.999 if precision == "bf16" else 0.99998  # fp16: 3 more significand bits
            assert _cos(img, want_i).min() > floor, _cos(img, want_i)
            assert _cos(txt, want_t).min() > floor, _cos(txt, want_t)
        if n_extra:  # a ragged batch larger than one GEMM tile, vs the oracle
            from oracle import oracle as orc

            o = orc.OracleCLIP(geo, synth_state_dict(geo, 0))
            px2, _ = make_pixels(n_extra, geo.image_size, 10, ood=True, seed=3)
            got = net.get_image_features(pixel_values=torch.from_numpy(px2).cuda(), normalize=True).cpu().numpy()
            want = o.encode_image(px2)
            if precision == "fp32":
                np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
            else:
                assert _cos(got, want).min() > 0.999
    finally:
        net.close()


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_get_ood_scores_clip_vs_reference_outputs(golden_dir, precision):
    """Drive the re-written hot function exactly as the reference was driven when the
    fixture was captured (same weights, pixels, token ids, batch size)."""
    from mcm_amd import detection
    from mcm_amd.metrics import get_measures

    g = np.load(os.path.join(golden_dir, "scores_tiny.npz"))
    K, n_id, n_ood, bs = int(g["K"]), int(g["n_id"]), int(g["n_ood"]), int(g["batch"])
    geo = geometry("tiny")
    net = _net("tiny", precision, max_batch=bs, max_prompt_tokens=2048)
    ids, mask = make_token_ids(K, seed=2)

    class FixedTok:  # the fixture was captured with these ids (no vocabulary offline)
        def __call__(self, texts, padding=True, return_tensors="pt"):
            assert len(texts) == K
            return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}

    old = detection.load_tokenizer
    detection.load_tokenizer = lambda ckpt, **kw: FixedTok()
    try:
        l_in = SyntheticLoader(SyntheticImageSet(n_id, geo.image_size, K, False, 1), bs)
        l_out = SyntheticLoader(SyntheticImageSet(n_ood, geo.image_size, K, True, 1), bs)
        for score in SCORE_KINDS:
            for T in (1, 2):
                args = types.SimpleNamespace(ckpt="ViT-B/16", model="CLIP", score=score, T=T)
                s_in = detection.get_ood_scores_clip(args, net, l_in, class_names(K), in_dist=True)
                s_out = detection.get_ood_scores_clip(args, net, l_out, class_names(K))
                assert s_in.dtype == np.float32 and s_in.shape == (n_id,)
                if precision == "fp32":
                    tol = dict(rtol=5e-5, atol=2e-6) if score != "var" else dict(rtol=5e-3, atol=1e-9)
                    np.testing.assert_allclose(s_in, g[f"{score}_T{T}_in"], **tol)
                    np.testing.assert_allclose(s_out, g[f"{score}_T{T}_out"], **tol)
                if score == "MCM":
                    got = np.array(get_measures(-s_in, -s_out))
                    want = g[f"measures_T{T}"]
                    if precision == "fp32":
                        np.testing.assert_allclose(got, want, atol=1e-4)
                    else:  # 88 samples: one rank swap moves AUROC by 5e-4; report, bound loosely
                        np.testing.assert_allclose(got, want, atol=2e-2)
                if precision == "fp16":  # the default dtype against the reference's own scores
                    tol = {"MCM": dict(rtol=2e-4, atol=1e-6), "max-logit": dict(rtol=0, atol=2e-4),
                           "energy": dict(rtol=2e-4, atol=2e-4), "entropy": dict(rtol=2e-4, atol=1e-5),
                           "var": dict(rtol=5e-2, atol=1e-8)}[score]
                    np.testing.assert_allclose(s_in, g[f"{score}_T{T}_in"], **tol)
                    np.testing.assert_allclose(s_out, g[f"{score}_T{T}_out"], **tol)
    finally:
        detection.load_tokenizer = old
        net.close()


def _auroc_case(name, K, n, precisions, fp16_exact_weights=False):
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.metrics import get_measures
    from oracle import oracle as orc

    geo = geometry(name)
    sd = synth_state_dict(geo, 0)
    if fp16_exact_weights:  # the reference's checkpoints: weights released in fp16, exact as fp16 MFMA operands
        sd = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    ids, _ = make_token_ids(K, seed=2)
    o = orc.OracleCLIP(geo, sd)
    txt_o = o.encode_text(ids)
    px_in, _ = make_pixels(n, geo.image_size, K, ood=False, seed=1)
    px_out, _ = make_pixels(n, geo.image_size, K, ood=True, seed=1)
    enc = lambda px: np.concatenate([o.encode_image(px[i:i + 2048]) for i in range(0, n, 2048)])  # noqa: E731
    want_in = orc.score_features(enc(px_in), txt_o, 1.0, 0)
    want_out = orc.score_features(enc(px_out), txt_o, 1.0, 0)
    want = np.array(get_measures(-want_in, -want_out))
    report = {}
    for precision in precisions:
        net = NativeCLIP(geo, sd, precision=precision, max_batch=256, max_prompt_tokens=2048)
        try:
            txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
            s_in = np.concatenate([net.score_images(torch.from_numpy(px_in[i:i + 4096]).cuda(), txt, 1.0, "MCM")
                                   .cpu().numpy() for i in range(0, n, 4096)])
            s_out = np.concatenate([net.score_images(torch.from_numpy(px_out[i:i + 4096]).cuda(), txt, 1.0, "MCM")
                                    .cpu().numpy() for i in range(0, n, 4096)])
        finally:
            net.close()
        got = np.array(get_measures(-s_in, -s_out))
        report[precision] = dict(d_auroc_aupr_fpr=np.abs(got - want), max_dscore=np.abs(s_in - want_in).max(),
                                 oracle=want)
    return report


@pytest.mark.parametrize("fp16_exact", [True, False], ids=["fp16-exact-weights", "fp32-valued-weights"])
def test_auroc_parity_vs_oracle_large_sample(fp16_exact):
    """North-star bar |ΔAUROC|, |ΔAUPR|, |ΔFPR95| ≤ 1e-4 vs the fp32 ORACLE (CPU), on a sample large enough that
    1e-4 is above the metric quantum of every metric (tiny geometry, 2 x 20 000 images: FPR95 moves in steps
    of 5e-5).  Both weight regimes are asserted:
      * fp16-exact — seeded weights rounded to fp16 for the oracle and every arm alike, the case of the reference's
        checkpoints (released in fp16): fp32 mode and fp16 mode (the benchmarked dtype) are held to the bar;
      * fp32-valued seeded weights — the fp16 arm additionally rounds its operand copies of the weights: AUROC and
        AUPR are held to the bar, FPR95 to the MEASURED bound of 3e-4 (round 2 measured 2e-4 = 4 of 20 000
        samples on this geometry; it does not meet 1e-4 here and the test says so instead of hiding the regime).
    bf16 is the documented coarser arm and is bounded so a regression shows."""
    rep = _auroc_case("tiny", K=20, n=20000, precisions=("fp32", "fp16", "bf16"), fp16_exact_weights=fp16_exact)
    print("tiny n=20000, fp16-exact weights =", fp16_exact, ":", rep)
    assert 0.05 < rep["fp32"]["oracle"][0] < 0.95  # non-degenerate AUROC
    assert rep["fp32"]["d_auroc_aupr_fpr"].max() <= 1e-4, rep
    d16 = rep["fp16"]["d_auroc_aupr_fpr"]
    if fp16_exact:
        assert d16.max() <= 1e-4, rep
    else:
        assert d16[0] <= 1e-4 and d16[1] <= 1e-4 and d16[2] <= 3e-4, rep
    d = rep["bf16"]["d_auroc_aupr_fpr"]
    assert d[0] <= 1e-3 and d[2] <= 5e-3, rep  # bf16 operands: measured drift, see DESIGN.md


def test_auroc_parity_vs_oracle_b16_2l():
    """Same on a 2-layer full-width B/16 (2x96 images: AUROC quantum 1.1e-4, FPR quantum 1e-2)."""
    rep = _auroc_case("B16-2L", K=20, n=96, precisions=("fp32", "bf16"))
    print("B16-2L n=96:", rep)
    assert rep["fp32"]["max_dscore"] < 1e-6
    assert rep["fp32"]["d_auroc_aupr_fpr"][0] <= 2.5e-4, rep   # ≤ 2 pair swaps
    assert rep["bf16"]["d_auroc_aupr_fpr"][0] <= 5e-3, rep


@pytest.fixture(scope="module", params=[("fp16", "fp16-exact"), ("fp16", "fp32"), ("bf16", "fp16-exact")],
                ids=["fp16-single-operand", "fp16-split-weights", "bf16"])
def b16(request):
    """Full-size B/16 handle: the default dtype (fp16) in both weight forms — one operand per weight on fp16-exact
    weights (the headline path), W_hi + W_lo on fp32-valued ones — and BASELINE's bf16."""
    prec, regime = request.param
    net = _net("ViT-B/16", prec, regime, max_batch=512, max_prompt_tokens=1000 * 20)
    net._regime = regime
    assert net.split_weights == (regime == "fp32" or prec == "bf16")  # (fp16 values are not bf16 numbers)
    yield net
    net.close()


def test_full_size_properties(b16):
    """BASELINE config sizes (B/16, batch 512, K=1000): properties that need no oracle."""
    K = 1000
    ids, _ = make_token_ids(K, seed=2)
    txt = b16.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
    assert txt.shape == (K, 512)
    assert torch.allclose(txt.norm(dim=1), torch.ones(K, device="cuda"), atol=1e-5)
    g = torch.Generator(device="cuda").manual_seed(11)
    px = torch.randn((512, 3, 224, 224), generator=g, device="cuda")
    s_full = b16.score_images(px, txt, 1.0, "MCM")
    assert s_full.shape == (512,) and torch.isfinite(s_full).all()
    # determinism: same launch twice → bitwise equal
    assert torch.equal(s_full, b16.score_images(px, txt, 1.0, "MCM"))
    # batch-split invariance: rows are independent, so any split gives bitwise-equal scores
    s_split = torch.cat([b16.score_images(px[:200], txt), b16.score_images(px[200:203], txt),
                         b16.score_images(px[203:], txt)])
    assert torch.equal(s_full, s_split)
    # MCM ∈ [-1, -1/K]; permuting the prompt bank leaves max-softmax unchanged up to the
    # fp32 summation order of the softmax denominator
    assert (s_full <= -1.0 / K + 1e-7).all() and (s_full >= -1.0).all()
    perm = torch.randperm(K, generator=torch.Generator().manual_seed(3)).cuda()
    s_perm = b16.score_images(px[:64], txt[perm], 1.0, "MCM")
    torch.testing.assert_close(s_perm, s_full[:64], rtol=1e-6, atol=1e-9)
    # feature path == fused path
    f = b16.get_image_features(pixel_values=px[:64], normalize=True)
    torch.testing.assert_close(b16.score_features(f, txt, 1.0, "MCM"), s_full[:64], rtol=0, atol=0)
    # max-logit is the plain cosine: bounded by 1 and equal to the max of f @ txt.T
    ml = b16.score_features(f, txt, 1.0, "max-logit")
    torch.testing.assert_close(-ml, (f @ txt.T).max(dim=1).values, rtol=1e-5, atol=1e-6)
    torch.cuda.synchronize()
    assert b16.kernel_faults == 0   # full-size runs through the persistent attention kernel: no wait ran out of its budget


def test_full_size_bf16_vs_oracle_small_sample(b16):
    """Full-depth B/16 bf16 features vs the fp32 oracle on 4 images (oracle: seconds)."""
    from oracle import oracle as orc

    geo = geometry("ViT-B/16")
    o = orc.OracleCLIP(geo, synth_state_dict(geo, 0, b16._regime))
    px, _ = make_pixels(4, 224, 10, ood=False, seed=9)
    want = o.encode_image(px)
    got = b16.get_image_features(pixel_values=torch.from_numpy(px).cuda(), normalize=True).cpu().numpy()
    c = _cos(got, want)
    print("16-bit vs fp32-oracle cosine (full B/16):", c, "max|d|", np.abs(got - want).max())
    assert c.min() > (0.999 if b16.precision == 0 else 0.99998)  # bf16 / fp16 (3 more significand bits)


def test_errors_match_reference_behaviour(b16):
    with pytest.raises(ValueError):  # HF modeling_clip.py:204-207
        b16.get_image_features(pixel_values=torch.zeros((1, 3, 200, 200), device="cuda"))
    with pytest.raises(ValueError):  # HF modeling_clip.py:241-245
        b16.get_text_features(input_ids=torch.full((1, 78), 49407))
    with pytest.raises(RuntimeError):
        b16.score_features(torch.zeros((1, 512), device="cuda"), torch.zeros((4, 512), device="cuda"),
                           T=0.0)


@pytest.mark.parametrize("name", ["ViT-B/32", "ViT-L/14"])
def test_other_checkpoints_vs_oracle(name):
    """The other two --CLIP_ckpt geometries (reference eval_ood_detection.py:34-35): B/32 (50
    tokens, 32-px patches) and L/14 (257 tokens, 14-px patches → K padded 588→640, width 1024,
    24 layers, proj 768), bf16 features vs the fp32 oracle on 2 images / 3 prompts."""
    from oracle import oracle as orc

    geo = geometry(name)
    sd = synth_state_dict(geo, 0)
    o = orc.OracleCLIP(geo, sd)
    px, _ = make_pixels(2, geo.image_size, 10, ood=False, seed=4)
    ids, _ = make_token_ids(3, seed=5)
    want_i, want_t = o.encode_image(px), o.encode_text(ids)
    net = _net(name, "bf16", max_batch=4, max_prompt_tokens=1024)
    try:
        got_i = net.get_image_features(pixel_values=torch.from_numpy(px).cuda()).cpu().numpy()
        got_t = net.get_text_features(input_ids=torch.from_numpy(ids)).cpu().numpy()
    finally:
        net.close()
    assert got_i.shape == (2, geo.proj_dim) and got_t.shape == (3, geo.proj_dim)
    assert _cos(got_i, want_i).min() > 0.999 and _cos(got_t, want_t).min() > 0.999


def test_uint8_ingest_matches_float_path():
    """N2: uint8 NHWC + fused ToTensor/Normalize == the fp32 NCHW path on the same pixels."""
    from oracle import oracle as orc

    geo = geometry("B16-2L")
    sd = synth_state_dict(geo, 0)
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, size=(5, geo.image_size, geo.image_size, 3), dtype=np.uint8)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    f32 = ((u8.astype(np.float32) / np.float32(255.0) - mean) / std).transpose(0, 3, 1, 2).copy()
    want = orc.OracleCLIP(geo, sd).encode_image(f32)
    net = _net("B16-2L", "fp32", max_batch=8, max_prompt_tokens=1024)
    try:
        got_u8 = net.get_image_features(pixel_values=torch.from_numpy(u8).cuda(), normalize=True).cpu().numpy()
        got_f = net.get_image_features(pixel_values=torch.from_numpy(f32).cuda(), normalize=True).cpu().numpy()
        raw_u8 = net.get_image_features(pixel_values=torch.from_numpy(u8).cuda()).cpu().numpy()
        np.testing.assert_allclose(_unit(raw_u8), got_u8, rtol=0, atol=1e-6)
    finally:
        net.close()
    np.testing.assert_allclose(got_u8, want, rtol=0, atol=2e-4)
    np.testing.assert_allclose(got_u8, got_f, rtol=0, atol=2e-6)


def test_prompt_ensemble_bank():
    """N3: bank[k] = normalise(mean_t normalise(feat[k,t])) vs numpy on the same features."""
    from mcm_amd import detection

    net = _net("tiny", "fp32", max_batch=8, max_prompt_tokens=4096)
    try:
        args = types.SimpleNamespace(ckpt="ViT-B/16", model="CLIP", score="MCM", T=1)
        labels = class_names(7)
        bank = detection.encode_prompt_ensemble(args, net, labels).cpu().numpy()
        T = len(detection.DEFAULT_TEMPLATES)
        tok = detection.load_tokenizer("x")
        prompts = [t.format(c=c) for c in labels for t in detection.DEFAULT_TEMPLATES]
        ids = tok(prompts, padding=True, return_tensors="pt")["input_ids"]
        feats = net.get_text_features(input_ids=ids, normalize=True).cpu().numpy().reshape(7, T, -1)
    finally:
        net.close()
    want = feats.mean(axis=1)
    want /= np.linalg.norm(want, axis=1, keepdims=True)
    assert bank.shape == (7, 64)
    np.testing.assert_allclose(bank, want, rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,batches", [("ViT-B/16", (512, 100, 57)), ("ViT-B/32", (512, 300)), ("ViT-L/14", (64,))])
@pytest.mark.parametrize("precision", ["fp16", "bf16", "fp32"])
def test_patch_gemm_gathering_pixels_equals_the_patchify_route(name, batches, precision):
    """SURVEY.md K1, the im2col-free patch embedding: since round 4 the patch GEMM reads its A operand from the fp32 NCHW
    pixels itself (gemm_p256_kernel<..., PXF>) wherever the persistent kernel takes the problem and the patch size divides
    a K-step (B/16, B/32 from batch 56 / 223 on).  Against the rounds 1 - 3 route (patchify writes a patch matrix, the GEMM
    reads it back; harness switch mcm_debug_patch_fold(0)): the same bits, whole and ragged batches, every dtype; and
    ViT-L/14 (P = 14, padded K) and small batches keep the old route."""
    from mcm_amd.engine import NativeCLIP

    geo = geometry(name)
    B = max(batches)
    net = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision=precision, max_batch=B, max_prompt_tokens=77,
                     harness=True)
    try:
        g = torch.Generator(device="cuda").manual_seed(17)
        px = torch.randn((B, 3, geo.image_size, geo.image_size), generator=g, device="cuda")
        for b in batches:
            assert net._lib.mcm_debug_patch_fold(1) == 0
            got = net.get_image_features(px[:b])
            assert net._lib.mcm_debug_patch_fold(0) == 0
            want = net.get_image_features(px[:b])
            assert torch.isfinite(got).all() and torch.equal(got, want), (name, precision, b)
    finally:
        net._lib.mcm_debug_patch_fold(1)
        net.close()
