"""LayerNorm by the row panel's cluster of workgroups (gemm_arms.hpp "LNC", harness switch mcm_debug_ln_cluster; round 6: the
full-row residual-GEMM epilogue VERDICT r5 item 2 asked to be BUILT).  The N / 256 workgroups that hold the tiles of one
256-row panel exchange 64-column row moments through their XCD's L2 and write the LayerNorm output from the accumulator
registers: no LayerNorm launch, no re-read of the residual stream.  Statistics are slot-wise two-pass moments combined by
Chan's formula — the same accuracy as the LayerNorm kernel's whole-row two-pass statistics in another summation order — so
scores are held to fp32 round-off of the run with every LayerNorm launched, to bitwise repeatability over a long series of
launches (the counters in device memory must be left zeroed), and no wait may ever have given up."""
import ctypes

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


def _timeouts(net):
    n = ctypes.c_uint64(0)
    assert net._lib.mcm_debug_ln_tail_timeouts(net._h, ctypes.byref(n)) == 0
    return n.value


@pytest.mark.parametrize("ckpt,precision,batch", [("ViT-B/16", "fp16", 160), ("ViT-B/16", "bf16", 512),
                                                   ("ViT-L/14", "fp16", 64), ("ViT-B/32", "fp16", 512)])
def test_ln_cluster_equals_the_layernorm_launches_to_round_off(ckpt, precision, batch):
    geo = geometry(ckpt)
    sd = _affine_state(geo)
    ids, _ = make_token_ids(40, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision=precision, max_batch=batch, max_prompt_tokens=40 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        g = torch.Generator(device="cuda").manual_seed(13)
        px = torch.randn((batch, 3, geo.image_size, geo.image_size), generator=g, device="cuda")
        assert net._lib.mcm_debug_ln_cluster(0) == 0
        launched = net.score_images(px, txt, 1.0, "MCM").clone()
        feats = net.get_image_features(pixel_values=px, normalize=True).clone()
        assert net._lib.mcm_debug_ln_cluster(1) == 0  # (an A/B arm: the shipped library launches its LayerNorms)
        first = None
        reps = 12 if batch <= 160 else 4
        for _ in range(reps):  # every launch must leave the counters zeroed for the next one
            got = net.score_images(px, txt, 1.0, "MCM").clone()
            first = got if first is None else first
            assert torch.equal(got, first)                       # deterministic: the Chan combination has a fixed order
        torch.cuda.synchronize()
        assert _timeouts(net) == 0 and net.kernel_faults == 0
        assert torch.isfinite(first).all()
        # against the launched LayerNorms: a LayerNorm output differs by an ulp of the 16-bit operand now and then (statistics
        # in another summation order), which moves a score by a fraction of the 16-bit arm's own noise
        f2 = net.get_image_features(pixel_values=px, normalize=True)
        cos = (f2 * feats).sum(dim=1)
        print(f"{ckpt} {precision} batch {batch}: max |d score| {float((first - launched).abs().max()):.3e} "
              f"(score ~{float(launched.abs().mean()):.3e}), min cos(features) {float(cos.min()):.8f}")
        tol = 2e-5 if precision == "fp16" else 2e-4
        assert float((first - launched).abs().max()) <= tol * float(launched.abs().max())
        assert float(cos.min()) > 1 - (1e-6 if precision == "fp16" else 1e-4)
        ragged = torch.cat([net.score_images(px[: batch // 2 + 3], txt).clone(),
                            net.score_images(px[batch // 2 + 3:], txt).clone()])
        # pad rows, other row-tile counts, the same state buffer.  (Not bitwise: a sub-batch too small for the ping-pong kernel
        # takes the LayerNorm LAUNCHES, whose statistics are summed in another order than the cluster's.)
        assert float((ragged - first).abs().max()) <= tol * float(launched.abs().max())
        assert _timeouts(net) == 0
    finally:
        net._lib.mcm_debug_ln_cluster(0)
        net.close()


def test_ln_cluster_in_a_long_run():
    """60 back-to-back batches (the bench's hot loop) with the cluster arm on: the same scores every time, no wait gave up."""
    geo = geometry("ViT-B/16")
    sd = _affine_state(geo)
    ids, _ = make_token_ids(40, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision="fp16", max_batch=512, max_prompt_tokens=40 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        assert net._lib.mcm_debug_ln_cluster(1) == 0
        g = torch.Generator(device="cuda").manual_seed(5)
        px = torch.randn((512, 3, 224, 224), generator=g, device="cuda")
        out = torch.empty((60, 512), device="cuda")
        for i in range(60):
            net.score_images(px, txt, 1.0, "MCM", out=out[i])
        torch.cuda.synchronize()
        assert torch.equal(out, out[0].expand_as(out))
        assert _timeouts(net) == 0 and net.kernel_faults == 0
    finally:
        net._lib.mcm_debug_ln_cluster(0)
        net.close()


def _deferred(net):
    n = ctypes.c_uint64(0)
    assert net._lib.mcm_debug_ln_cluster_deferred(net._h, ctypes.byref(n)) == 0
    return n.value


@pytest.mark.parametrize("ckpt,precision,batch", [("ViT-B/16", "fp16", 512), ("ViT-B/16", "bf16", 160), ("ViT-L/14", "fp16", 64)])
def test_ln_cluster_defer_form_same_bits_as_the_waiting_form(ckpt, precision, batch):
    """The defer form (mcm_debug_ln_cluster_spin >= 0, R6.4): a wave whose partners have not published within the poll budget
    leaves its 128 x 64 segment to the clean-up launch, which normalises it from x and the slot moments with the in-kernel
    arithmetic.  Which segments are deferred depends on timing, the bits must not: budgets 0 (defer whatever is not ready at once),
    4 and 64 against the waiting form, repeated; with budget 0 segments ARE deferred (otherwise the clean-up path went untested);
    the masks are left zeroed (a later waiting-form run and the LayerNorm launches still agree)."""
    geo = geometry(ckpt)
    sd = _affine_state(geo)
    ids, _ = make_token_ids(40, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision=precision, max_batch=batch, max_prompt_tokens=40 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        g = torch.Generator(device="cuda").manual_seed(17)
        px = torch.randn((batch, 3, geo.image_size, geo.image_size), generator=g, device="cuda")
        assert net._lib.mcm_debug_ln_cluster(1) == 0
        assert net._lib.mcm_debug_ln_cluster_spin(-1) == 0
        waited = net.score_images(px, txt, 1.0, "MCM").clone()
        assert _deferred(net) == 0
        seen = {}
        for budget in (0, 4, 64, 0):
            assert net._lib.mcm_debug_ln_cluster_spin(budget) == 0
            before = _deferred(net)
            for _ in range(3):
                got = net.score_images(px, txt, 1.0, "MCM").clone()
                assert torch.equal(got, waited), f"budget {budget}"
            seen[budget] = _deferred(net) - before
        print(f"{ckpt} {precision} batch {batch}: segments deferred per 3 passes by poll budget {seen}")
        assert seen[0] > 0
        assert net._lib.mcm_debug_ln_cluster_spin(-1) == 0
        assert torch.equal(net.score_images(px, txt, 1.0, "MCM"), waited)
        torch.cuda.synchronize()
        assert _timeouts(net) == 0 and net.kernel_faults == 0
    finally:
        net._lib.mcm_debug_ln_cluster_spin(-1)
        net._lib.mcm_debug_ln_cluster(0)
        net.close()
