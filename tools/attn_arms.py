"""attn_arms.py — the round-5 arms of the 16-bit attention kernel at the headline shape (B/16, batch 512, fp16), through the harness
library: waves per workgroup (8 shipped, 7, 6, 5) and the phase probes (EXPERIMENTS.md R5.9: reads from the L2 / no arithmetic / no
stores — wrong results by design, times only).  Usage (GPU box): python tools/attn_arms.py [iters]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="fp16", max_batch=8, max_prompt_tokens=2048, harness=True)
lib = net._lib
nseq, L, heads = 512, 197, 12
D = heads * 64
g = torch.Generator(device="cuda").manual_seed(L)
qkv = torch.randn((nseq * L, 3 * D), device="cuda", generator=g)
qkv[:, :2 * D] *= 1.5
qkv = qkv.half()
# a second buffer the size of the Infinity Cache is written between timed batches so that every arm starts from the same (cold) state
names = {1: "8 waves (shipped)", 21: "persistent, 4 loaders, window 4", 24: "persistent, 2 loaders, window 8",
         32: "persistent K/V units, 2 loaders, window 8", 33: "persistent K/V units, 2 loaders, window 4",
         34: "persistent K/V units, 2 loaders", 35: "persistent K/V units, 1 loader, window 8",
         30: "persistent probe: 2 loaders w8, compute never waits"}
PROBES = (15, 16, 17, 25, 26, 27, 30, 31)
outs, times = {}, {v: [] for v in names}
for rnd in range(3):
    for v in names:
        assert lib.mcm_debug_attention_variant(v) == 0
        out = torch.zeros((nseq * L, D), device="cuda", dtype=torch.float16)

        def f():
            rc = lib.mcm_op_attention(net._h, 2, ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), nseq, L, heads, 0, None)
            assert rc == 0, lib.mcm_last_error(net._h)

        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) * 1e3 / iters)
        outs[v] = out
# sustained: ~1.5 s per variant with the clock and package power sampled (rocm-smi): which arms run at which clock
from tools.bench_legs import SmiSampler  # noqa: E402
sustained = {}
for v in names:
    assert lib.mcm_debug_attention_variant(v) == 0
    out = torch.zeros((nseq * L, D), device="cuda", dtype=torch.float16)
    smi = SmiSampler(period=0.1)
    smi.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_it = 8000
    e0.record()
    for _ in range(n_it):
        lib.mcm_op_attention(net._h, 2, ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), nseq, L, heads, 0, None)
    e1.record()
    torch.cuda.synchronize()
    r = smi.stop()
    sustained[v] = (e0.elapsed_time(e1) * 1e3 / n_it, r.get("sclk_mhz_mean"), r.get("power_w_mean"), r.get("busy_samples"))
for v, n in names.items():
    eq = "" if v in PROBES else f"   bit-equal to shipped: {torch.equal(outs[v], outs[1])}"
    print(f"variant {v:2d} {n:40s} " + " / ".join(f"{t:6.1f}" for t in times[v]) + " us" + eq
          + "   sustained %.1f us, sclk %s MHz, %s W (%s samples)" % sustained[v], flush=True)
lib.mcm_debug_attention_variant(1)
net.close()
