"""A/B of the two 16-bit attention kernels through the C ABI (mcm_op_attention): time per launch at the
model's shapes and max |difference| between the variants and against a float64 torch reference.
Usage: python tools/attn_probe.py [iters]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="bf16", max_batch=8, max_prompt_tokens=2048, harness=True)
lib = net._lib


def ref(qkv, nseq, L, heads, causal):
    D = heads * 64
    x = qkv.double().view(nseq, L, 3, heads, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=s.device, dtype=s.dtype).triu(1)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(nseq * L, D)


for prec, dt in ((0, torch.bfloat16), (2, torch.float16)):
    for nseq, L, heads, causal in ((512, 197, 12, 0), (256, 257, 16, 0), (512, 50, 12, 0), (1000, 16, 8, 1),
                                   (200, 77, 8, 1)):
        D = heads * 64
        g = torch.Generator(device="cuda").manual_seed(L)
        qkv = torch.randn((nseq * L, 3 * D), device="cuda", generator=g)
        qkv[:, :2 * D] *= 1.5
        qkv = qkv.to(dt)
        outs = []
        for variant in ((0, 1, 10, 1, 10) if L in (197, 257) else (0, 1, 10)):
            assert lib.mcm_debug_attention_variant(variant) == 0
            out = torch.zeros((nseq * L, D), device="cuda", dtype=dt)

            def f():
                rc = lib.mcm_op_attention(net._h, prec, ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                          nseq, L, heads, causal, None)
                assert rc == 0, lib.mcm_last_error(net._h)

            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            outs.append(out.float())
            print(f"prec={prec} nseq={nseq} L={L} heads={heads} causal={causal} variant={variant}: {us:8.1f} us  "
                  f"({(qkv.numel() + out.numel()) * 2 / us / 1e6:.2f} TB/s)", flush=True)
        n_ref = min(nseq, 8)
        want = ref(qkv[: n_ref * L], n_ref, L, heads, causal).float()
        print("    max|v-ref| per variant: " + "  ".join(f"{(o[: n_ref * L] - want).abs().max().item():.3e}" for o in outs)
              + "   all new-kernel variants bit-equal: " + str(all(torch.equal(outs[1], o) for o in outs[2:])), flush=True)
lib.mcm_debug_attention_variant(1)
net.close()
