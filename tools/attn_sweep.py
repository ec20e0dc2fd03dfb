"""attn_sweep.py — shipped 8-wave attention kernel vs the persistent form over batch sizes (B/16 shape, fp16 and bf16): time per launch
and bit-equality.  Usage (GPU box): python tools/attn_sweep.py [variant]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

var = int(sys.argv[1]) if len(sys.argv) > 1 else 21
geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="fp16", max_batch=8, max_prompt_tokens=2048, harness=True)
lib = net._lib
heads = 12
D = heads * 64
for prec, dt in ((2, torch.float16), (0, torch.bfloat16)):
    for L in (197, 193, 208):
        for nseq in (1, 7, 22, 43, 44, 64, 100, 128, 256, 512, 768):
            if (L != 197 or prec == 0) and nseq not in (7, 43, 100, 512):
                continue
            g = torch.Generator(device="cuda").manual_seed(L + nseq)
            qkv = (torch.randn((nseq * L, 3 * D), device="cuda", generator=g) * 1.5).to(dt)
            res = {}
            for v in (1, var):
                assert lib.mcm_debug_attention_variant(v) == 0
                out = torch.zeros((nseq * L, D), device="cuda", dtype=dt)

                def f():
                    rc = lib.mcm_op_attention(net._h, prec, ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), nseq, L, heads, 0, None)
                    assert rc == 0, lib.mcm_last_error(net._h)

                for _ in range(5):
                    f()
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(40):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 40)
                res[v] = (min(ts), out)
            print(f"prec={prec} L={L} nseq={nseq:4d} jobs={nseq * heads:5d}: shipped {res[1][0]:7.1f} us, variant {var} {res[var][0]:7.1f} us "
                  f"({res[var][0] / res[1][0]:.3f})  bit-equal {torch.equal(res[1][1], res[var][1])}", flush=True)
lib.mcm_debug_attention_variant(1)
net.close()
