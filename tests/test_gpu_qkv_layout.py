"""qkv head-major between the QKV projection and attention (GemmArgs::hm, launch_attention's hm; an A/B arm of the
harness library, mcm_debug_qkv_head_major — DESIGN.md 5.5): the same numbers in another order in h->qkv, so scores
must be bit-identical to the shipped [rows][3 D] form, for whole and ragged batches (ping-pong and tile kernels) and
with the LayerNorm fold's consumer epilogue in front of the same store."""
import pytest
import torch

from mcm_amd.config import geometry
from mcm_amd.engine import NativeCLIP
from mcm_amd.synth import make_token_ids
from mcm_amd.weights import synth_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ckpt,precision,batch", [("ViT-B/16", "fp16", 160), ("ViT-B/16", "bf16", 37),
                                                   ("ViT-B/32", "fp16", 96), ("B16-2L", "fp16", 5)])
def test_head_major_qkv_is_bit_identical(ckpt, precision, batch):
    geo = geometry(ckpt)
    sd = synth_state_dict(geo, 0)
    ids, _ = make_token_ids(40, seed=2)
    net = NativeCLIP(geo, sd, device=0, precision=precision, max_batch=batch, max_prompt_tokens=40 * 20, harness=True, weight_operands="single")
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        g = torch.Generator(device="cuda").manual_seed(11)
        px = torch.randn((batch, 3, geo.image_size, geo.image_size), generator=g, device="cuda")
        row = net.score_images(px, txt, 1.0, "MCM").clone()
        assert net._lib.mcm_debug_qkv_head_major(1) == 0
        head = net.score_images(px, txt, 1.0, "MCM").clone()
        part = torch.cat([net.score_images(px[: batch // 2], txt).clone(), net.score_images(px[batch // 2:], txt).clone()])
        assert torch.isfinite(row).all()
        assert torch.equal(head, row) and torch.equal(part, row)
        if ckpt == "ViT-B/16" and precision == "fp16":  # with the LayerNorm fold's consumer epilogue
            assert net._lib.mcm_debug_ln_fold(1) == 0
            fold_head = net.score_images(px, txt, 1.0, "MCM").clone()
            assert net._lib.mcm_debug_qkv_head_major(0) == 0
            assert torch.equal(fold_head, net.score_images(px, txt, 1.0, "MCM"))
    finally:
        net._lib.mcm_debug_ln_fold(0)
        net._lib.mcm_debug_qkv_head_major(0)
        net.close()
