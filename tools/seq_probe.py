"""Does a consumer run faster when its producer's stores were allowed to stay in the caches?  QKV GEMM (variant:
0 = tile kernel with plain stores, 5 = ping-pong kernel with streamed `nt` stores) -> attention, and
fc1 -> fc2, each consumer timed with its own events inside the alternating sequence.
Usage: python tools/seq_probe.py [iters]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prec, dt = 2, torch.float16
geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="fp16", max_batch=8, max_prompt_tokens=2048, harness=True)
lib = net._lib
B, L, H, D, F = 512, 197, 12, 768, 3072
M = B * L
g = torch.Generator(device="cuda").manual_seed(1)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
x = torch.randn((M, D), device="cuda", generator=g).to(dt)
wq = (torch.randn((3 * D, D), device="cuda", generator=g) * D ** -0.5).to(dt)
w1 = (torch.randn((F, D), device="cuda", generator=g) * D ** -0.5).to(dt)
w2 = (torch.randn((D, F), device="cuda", generator=g) * F ** -0.5).to(dt)
bq, b1, b2 = (0.1 * torch.randn(n, device="cuda", generator=g) for n in (3 * D, F, D))
qkv = torch.zeros((M, 3 * D), device="cuda", dtype=dt)
att = torch.zeros((M, D), device="cuda", dtype=dt)
h = torch.zeros((M, F), device="cuda", dtype=dt)
resid = torch.randn((M, D), device="cuda", generator=g)


def lin(xx, w, b, y, N, K, epi):
    rc = lib.mcm_op_linear(net._h, prec, p(xx), p(w), p(b), p(y), p(resid), M, N, K, epi, None)
    assert rc == 0, lib.mcm_last_error(net._h)


def attention():
    rc = lib.mcm_op_attention(net._h, prec, p(qkv), p(att), B, L, H, 0, None)
    assert rc == 0, lib.mcm_last_error(net._h)


def timed(producer, consumer):
    for _ in range(2):
        producer(); consumer()
    torch.cuda.synchronize()
    tp, tc = [], []
    for _ in range(iters):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); producer(); e[1].record(); consumer(); e[2].record()
        tp.append((e[0], e[1])); tc.append((e[1], e[2]))
    torch.cuda.synchronize()
    med = lambda ev: sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[len(ev) // 2]  # noqa: E731
    return med(tp), med(tc)


for v in (0, 3, 5):
    lib.mcm_debug_gemm_variant(v)
    a, b = timed(lambda: lin(x, wq, bq, qkv, 3 * D, D, 0), attention)
    print(f"QKV variant {v}: {a:7.1f} us -> attention {b:7.1f} us   sum {a + b:7.1f}", flush=True)
for v in (0, 3, 5):
    def f1():
        lib.mcm_debug_gemm_variant(v)
        lin(x, w1, b1, h, F, D, 1)

    def f2():
        lib.mcm_debug_gemm_variant(5)
        lin(h, w2, b2, h, D, F, 2)
    a, b = timed(f1, f2)
    print(f"fc1 variant {v}: {a:7.1f} us -> fc2 (ping-pong) {b:7.1f} us   sum {a + b:7.1f}", flush=True)
lib.mcm_debug_gemm_variant(-1)
net.close()
