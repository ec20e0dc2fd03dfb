"""The LITERAL full-row tile (gemm_arms.hpp "ROW64", harness switch mcm_debug_ln_row; EXPERIMENTS.md R6.7): out-proj / fc2 of a
whole-batch 16-bit vision layer as 64-row x N tiles in one workgroup, whose epilogue writes the residual stream once and the
LayerNorm output from the accumulator registers — no LayerNorm launch, no re-read of x, no cross-workgroup traffic.  An A/B
arm built to be measured ("build, not cost on paper").  The GEMM sums K in the shipped kernels' order
(16x16x32 MFMAs over ascending K), the row statistics are two-pass like the LayerNorm kernel's in another summation order:
scores are held to fp32 round-off of the shipped path, and to bitwise repeatability."""
import numpy as np
import pytest
import torch

from mcm_amd.config import geometry
from mcm_amd.engine import NativeCLIP
from mcm_amd.synth import make_token_ids
from mcm_amd.weights import synth_state_dict

pytestmark = pytest.mark.gpu


def _affine_state(geo):
    sd = synth_state_dict(geo, 0)
    rng = np.random.default_rng(7)  # LayerNorm weights away from HF's (1, 0)
    for k in list(sd):
        if ".layer_norm" in k and k.startswith("vision_model"):
            sd[k] = ((1.0 if k.endswith(".weight") else 0.0) + 0.3 * rng.standard_normal(sd[k].shape)).astype(np.float32)
    return sd


# mcm_debug_ln_row(1 / 2): W one / two K-steps ahead; 3 / 4: the same with W in the blocked (piece-contiguous) layout
@pytest.mark.parametrize("stages", [1, 2, 3, 4], ids=["2-stage", "3-stage", "2-stage-blockedW", "3-stage-blockedW"])
@pytest.mark.parametrize("ckpt,precision,batch", [("ViT-B/16", "fp16", 160), ("ViT-B/16", "bf16", 512),
                                                   ("ViT-L/14", "fp16", 64), ("ViT-B/32", "fp16", 512), ("ViT-B/16", "fp16", 3)])
def test_full_row_tiles_equal_the_launched_layernorms_to_round_off(ckpt, precision, batch, stages):
    if stages in (2, 4) and ckpt == "ViT-L/14":
        pytest.skip("three W stages need N = 768 (160 KiB of LDS); N = 1024 runs two")
    geo = geometry(ckpt)
    sd = _affine_state(geo)
    ids, _ = make_token_ids(40, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision=precision, max_batch=batch, max_prompt_tokens=40 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        g = torch.Generator(device="cuda").manual_seed(13)
        px = torch.randn((batch, 3, geo.image_size, geo.image_size), generator=g, device="cuda")
        assert net._lib.mcm_debug_ln_row(0) == 0
        launched = net.score_images(px, txt, 1.0, "MCM").clone()
        feats = net.get_image_features(pixel_values=px, normalize=True).clone()
        assert net._lib.mcm_debug_ln_row(stages) == 0  # (an A/B arm: the shipped library launches its LayerNorms)
        first = None
        for _ in range(4):
            got = net.score_images(px, txt, 1.0, "MCM").clone()
            first = got if first is None else first
            assert torch.equal(got, first)   # fixed summation orders: deterministic
        torch.cuda.synchronize()
        assert torch.isfinite(first).all() and net.kernel_faults == 0
        f2 = net.get_image_features(pixel_values=px, normalize=True)
        cos = (f2 * feats).sum(dim=1)
        print(f"{ckpt} {precision} batch {batch}: max |d score| {float((first - launched).abs().max()):.3e} "
              f"(score ~{float(launched.abs().mean()):.3e}), min cos(features) {float(cos.min()):.8f}")
        tol = 2e-5 if precision == "fp16" else 2e-4
        assert float((first - launched).abs().max()) <= tol * float(launched.abs().max())
        assert float(cos.min()) > 1 - (1e-6 if precision == "fp16" else 1e-4)
        if batch > 8:   # ragged sub-batches: pad rows, other tile counts
            ragged = torch.cat([net.score_images(px[: batch // 2 + 3], txt).clone(), net.score_images(px[batch // 2 + 3:], txt).clone()])
            assert float((ragged - first).abs().max()) <= tol * float(launched.abs().max())
    finally:
        net._lib.mcm_debug_ln_row(0)
        net.close()
