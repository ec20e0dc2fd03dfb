"""LayerNorm fold (gemm.hip / mcm_api.hip run_layers), an A/B arm of the harness library (mcm_debug_ln_fold; measured 1 %
slower end to end than the LayerNorm launches, so the shipped library does not take it): in the 16-bit modes the
LayerNorms of the vision tower between a residual GEMM and the GEMM that consumes their output are not launched — the
residual epilogue writes gamma o x and the row moments, the consumer's epilogue normalises.  Checked here:

  * the folded tower against the same tower with every LayerNorm launched (harness switch mcm_debug_ln_fold) and against
    the exact-fp32 arm: the fold moves a rounding point, it must not move the result by more than that rounding;
  * the two producer forms (fused epilogue of the ping-pong kernel; plain residual GEMM + fold_rows_kernel for batches the
    tile kernel takes) and the two consumer forms give the SAME bits — a score does not depend on the batch it was in;
  * non-trivial gamma / beta (the seeded generator follows HF's init: gamma = 1, beta = 0, which would hide a wrong c or b').
"""
import numpy as np
import pytest
import torch

from mcm_amd.config import geometry
from mcm_amd.engine import NativeCLIP
from mcm_amd.synth import make_token_ids
from mcm_amd.weights import synth_state_dict

pytestmark = pytest.mark.gpu


def _state(geo, affine: bool):
    sd = synth_state_dict(geo, 0)
    if affine:  # LayerNorm weights away from (1, 0): every folded LayerNorm gets its own gamma / beta
        rng = np.random.default_rng(7)
        for k in list(sd):
            if ".layer_norm" in k and k.startswith("vision_model"):
                if k.endswith(".weight"):
                    sd[k] = (1.0 + 0.3 * rng.standard_normal(sd[k].shape)).astype(np.float32)
                else:
                    sd[k] = (0.2 * rng.standard_normal(sd[k].shape)).astype(np.float32)
    return sd


def _features(net, px, fold=None):
    if fold is not None:
        assert net._lib.mcm_debug_ln_fold(int(fold)) == 0
    try:
        return net.get_image_features(pixel_values=px, normalize=True).clone()
    finally:
        if fold is not None:
            net._lib.mcm_debug_ln_fold(0)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("affine", [False, True], ids=["hf-init-ln", "random-ln"])
def test_folded_tower_vs_unfolded_and_fp32(prec, affine):
    geo = geometry("ViT-B/16")
    sd = _state(geo, affine)
    g = torch.Generator(device="cuda").manual_seed(5)
    px = torch.randn((128, 3, 224, 224), generator=g, device="cuda")
    ref = NativeCLIP(geo, sd, device=0, precision="fp32", max_batch=128, max_prompt_tokens=77)
    try:
        want = ref.get_image_features(pixel_values=px, normalize=True).clone()
    finally:
        ref.close()
    net = NativeCLIP(geo, sd, device=0, precision=prec, max_batch=128, max_prompt_tokens=77, harness=True, weight_operands="single")
    try:
        on = _features(net, px, fold=True)     # 128 images: 99 M tiles x 3 > half a round -> ping-pong kernel, fused epilogues
        off = _features(net, px, fold=False)
        assert not torch.equal(on, off), "the switch did nothing: the fold path was not taken"
        small_on = _features(net, px[:5], fold=True)   # 5 images: tile kernel, fold_rows producer, tile consumer
        small_off = _features(net, px[:5], fold=False)
        assert torch.equal(small_on, on[:5]), "producer / consumer forms of the fold disagree"
        assert torch.equal(small_off, off[:5])
        assert net.saturation_count() == 0
    finally:
        net.close()
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a, b, dim=1)
    c_on, c_off = cos(on, want), cos(off, want)
    print(f"{prec} affine={affine}: 1-cos vs fp32  folded {float((1 - c_on).max()):.3e}  unfolded {float((1 - c_off).max()):.3e}"
          f"  folded-vs-unfolded {float((1 - cos(on, off)).max()):.3e}")
    bound = 2e-5 if prec == "fp16" else 1e-3
    assert float((1 - c_on).max()) < bound
    # the fold must not be a worse approximation of the fp32 tower than the LayerNorm launches are (25 % slack: both
    # are sums of the same kind of rounding noise)
    # (fp16: both means are 2 - 4e-8, the round-off of a float32 cosine itself — half an ulp of 1.0 is 3e-8 — so the slack has
    # an absolute part of that size; bf16: 1e-5, where the relative part is the test)
    assert float((1 - c_on).mean()) < 1.25 * float((1 - c_off).mean()) + 3e-8


def test_fold_batch_invariance_through_the_score_call():
    """Ragged and whole batches, fused and unfused producer, ping-pong and tile consumers: bitwise equal scores."""
    geo = geometry("ViT-B/16")
    sd = _state(geo, True)
    ids, _ = make_token_ids(50, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision="fp16", max_batch=160, max_prompt_tokens=50 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        g = torch.Generator(device="cuda").manual_seed(6)
        px = torch.randn((160, 3, 224, 224), generator=g, device="cuda")
        plain = net.score_images(px, txt, 1.0, "MCM").clone()
        assert net._lib.mcm_debug_ln_fold(1) == 0
        full = net.score_images(px, txt, 1.0, "MCM").clone()
        parts = torch.cat([net.score_images(px[:97], txt).clone(), net.score_images(px[97:99], txt).clone(),
                           net.score_images(px[99:], txt).clone()])
        assert torch.equal(full, parts)
        assert torch.isfinite(full).all() and not torch.equal(full, plain)
        torch.testing.assert_close(full, plain, rtol=2e-4, atol=0)
    finally:
        net._lib.mcm_debug_ln_fold(0)
        net.close()
