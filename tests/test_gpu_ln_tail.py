"""LayerNorm in the tail of the residual GEMMs (gemm.hip "LayerNorm in the tail"; mcm_api.hip run_layers): the LayerNorm
behind a whole-batch out-proj / fc2 of a 16-bit vision tower is computed by that GEMM's own waves once they have run
out of tiles — same arithmetic as the LayerNorm kernel (ln_row.hpp), so scores must be bit-identical to the run with
every LayerNorm launched (harness switch mcm_debug_ln_tail), for every launch of a long series (the tail is driven by
counters in device memory that every launch must leave zeroed), and no ticket may ever have timed out."""
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
def test_ln_tail_is_bit_identical_to_the_layernorm_launches(ckpt, precision, batch):
    geo = geometry(ckpt)
    sd = _affine_state(geo)
    ids, _ = make_token_ids(40, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision=precision, max_batch=batch, max_prompt_tokens=40 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        g = torch.Generator(device="cuda").manual_seed(13)
        px = torch.randn((batch, 3, geo.image_size, geo.image_size), generator=g, device="cuda")
        assert net._lib.mcm_debug_ln_tail(0) == 0
        launched = net.score_images(px, txt, 1.0, "MCM").clone()
        assert net._lib.mcm_debug_ln_tail(1) == 0  # (an A/B arm: the shipped library launches its LayerNorms)
        reps = 12 if batch <= 160 else 4
        for _ in range(reps):  # every launch must leave the counters zeroed for the next one
            tail = net.score_images(px, txt, 1.0, "MCM").clone()
            assert torch.equal(tail, launched)
        ragged = torch.cat([net.score_images(px[: batch // 2 + 3], txt).clone(),
                            net.score_images(px[batch // 2 + 3:], txt).clone()])
        assert torch.equal(ragged, launched)  # pad rows, other row-tile counts, the same state buffer
        assert torch.isfinite(tail).all()
        assert _timeouts(net) == 0
    finally:
        net._lib.mcm_debug_ln_tail(0)
        net.close()


def test_ln_tail_in_a_long_run_against_the_fp32_arm():
    """60 back-to-back batches (the bench's hot loop) with the tail on: same scores every time, and as close to the
    exact-fp32 arm as the LayerNorm launches are (it is the same arithmetic)."""
    geo = geometry("ViT-B/16")
    sd = _affine_state(geo)
    ids, _ = make_token_ids(40, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision="fp16", max_batch=512, max_prompt_tokens=40 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        assert net._lib.mcm_debug_ln_tail(1) == 0
        g = torch.Generator(device="cuda").manual_seed(5)
        px = torch.randn((512, 3, 224, 224), generator=g, device="cuda")
        out = torch.empty((60, 512), device="cuda")
        for i in range(60):
            net.score_images(px, txt, 1.0, "MCM", out=out[i])
        torch.cuda.synchronize()
        assert torch.equal(out, out[0].expand_as(out))
        assert _timeouts(net) == 0
    finally:
        net._lib.mcm_debug_ln_tail(0)
        net.close()
