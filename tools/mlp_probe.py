"""GEMM kernel variants timed in the model's context, through the C ABI (mcm_op_linear): the MLP pair
fc1 (QuickGELU epilogue) -> fc2 (fp32 residual epilogue) and the attention-side pair QKV -> out-proj at the
B/16 batch-512 shapes, each kernel timed with its own events (a) repeated alone, (b) inside the alternating
sequence the tower runs.  Usage: python tools/mlp_probe.py [iters] [prec: fp16|bf16] [variants...]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
prec_name = sys.argv[2] if len(sys.argv) > 2 else "fp16"
variants = [int(v) for v in sys.argv[3:]] or [3, 5]
prec, dt = {"bf16": (0, torch.bfloat16), "fp16": (2, torch.float16)}[prec_name]
geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision=prec_name, max_batch=8, max_prompt_tokens=2048, harness=True)
lib = net._lib
M, D, F = 512 * 197, 768, 3072
g = torch.Generator(device="cuda").manual_seed(1)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
x = torch.randn((M, D), device="cuda", generator=g).to(dt)
w1 = (torch.randn((F, D), device="cuda", generator=g) * D ** -0.5).to(dt)
w2 = (torch.randn((D, F), device="cuda", generator=g) * F ** -0.5).to(dt)
wq = (torch.randn((3 * D, D), device="cuda", generator=g) * D ** -0.5).to(dt)
wo = (torch.randn((D, D), device="cuda", generator=g) * D ** -0.5).to(dt)
b1, b2, bq, bo = (0.1 * torch.randn(n, device="cuda", generator=g) for n in (F, D, 3 * D, D))
h = torch.zeros((M, F), device="cuda", dtype=dt)
qkv = torch.zeros((M, 3 * D), device="cuda", dtype=dt)
resid = torch.randn((M, D), device="cuda", generator=g)


def lin(xx, w, b, y, r, N, K, epi):
    rc = lib.mcm_op_linear(net._h, prec, p(xx), p(w), p(b), p(y), p(r), M, N, K, epi, None)
    assert rc == 0, lib.mcm_last_error(net._h)


OPS = {"fc1": lambda: lin(x, w1, b1, h, resid, F, D, 1), "fc2": lambda: lin(h, w2, b2, h, resid, D, F, 2),
       "qkv": lambda: lin(x, wq, bq, qkv, resid, 3 * D, D, 0), "out": lambda: lin(x, wo, bo, x, resid, D, D, 2)}
FLOP = {"fc1": 2.0 * M * F * D, "fc2": 2.0 * M * F * D, "qkv": 2.0 * M * 3 * D * D, "out": 2.0 * M * D * D}


def timed(seq, which):
    """run `seq` (list of op names) iters times; events around the ops named `which`"""
    for _ in range(2):
        for n in seq:
            OPS[n]()
    torch.cuda.synchronize()
    ev = []
    for _ in range(iters):
        for n in seq:
            if n == which:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                OPS[n]()
                e1.record()
                ev.append((e0, e1))
            else:
                OPS[n]()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


for v in variants:
    assert lib.mcm_debug_gemm_variant(v) == 0
    for name, seq in (("fc1", ["fc1"]), ("fc2", ["fc2"]), ("fc1", ["fc1", "fc2"]), ("fc2", ["fc1", "fc2"]),
                      ("qkv", ["qkv"]), ("out", ["out"]), ("qkv", ["qkv", "out"]), ("out", ["qkv", "out"])):
        us = timed(seq, name)
        print(f"variant {v} {prec_name} {name:4s} in {'+'.join(seq):8s}: {us:7.1f} us  {FLOP[name] / us / 1e6:7.1f} TFLOP/s", flush=True)
lib.mcm_debug_gemm_variant(-1)
net.close()
