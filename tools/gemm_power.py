"""gemm_power.py — which clock and package power each GEMM shape of the B/16 step sustains on its own (EXPERIMENTS.md R5.10): the four
layer GEMMs with their fused epilogues, and out-proj / fc2 once more with a plain 16-bit store instead of the fp32 residual
read-modify-write (what the epilogue costs), ~1.5 s of back-to-back launches each, rocm-smi sampled every 0.1 s.
Usage (GPU box): python tools/gemm_power.py"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402
from tools.bench_legs import SmiSampler  # noqa: E402

geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="fp16", max_batch=8, max_prompt_tokens=2048)
lib = net._lib
M = 512 * 197
g = torch.Generator(device="cuda").manual_seed(1)
CASES = [("QKV  (2304, 768)  +bias, 16-bit store", 2304, 768, 0), ("fc1  (3072, 768)  +bias, QuickGELU, 16-bit store", 3072, 768, 1),
         ("fc2  (768, 3072)  +bias, fp32 residual RMW", 768, 3072, 2), ("fc2  (768, 3072)  +bias, 16-bit store", 768, 3072, 0),
         ("out-proj (768, 768)  +bias, fp32 residual RMW", 768, 768, 2), ("out-proj (768, 768)  +bias, 16-bit store", 768, 768, 0)]
for name, N, K, epi in CASES:
    x = (torch.randn((M, K), device="cuda", generator=g) * 0.5).half()
    w = (torch.randn((N, K), device="cuda", generator=g) * 0.03).half()
    b = torch.zeros(N, device="cuda")
    y = torch.zeros((M, N), device="cuda", dtype=torch.float16)
    r = torch.zeros((M, N), device="cuda", dtype=torch.float32) if epi == 2 else None

    def f():
        rc = lib.mcm_op_linear(net._h, 2, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                               ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(r.data_ptr()) if r is not None else None, M, N, K, epi, None)
        assert rc == 0, lib.mcm_last_error(net._h)

    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us20 = e0.elapsed_time(e1) * 1e3 / 20
    n_it = max(200, int(2.5e6 / us20))
    smi = SmiSampler(period=0.1)
    smi.start()
    e0.record()
    for _ in range(n_it):
        f()
    e1.record()
    torch.cuda.synchronize()
    s = smi.stop()
    us = e0.elapsed_time(e1) * 1e3 / n_it
    tf = 2.0 * M * N * K / us / 1e6
    mb = (M * K * 2 + N * K * 2 + (M * N * 8 if epi == 2 else M * N * 2)) / 1e6
    print(f"{name:52s} (20 launches: {us20:7.1f} us) {us:7.1f} us  {tf:6.0f} TF/s  {mb / us:5.2f} TB/s algorithmic   sclk {s.get('sclk_mhz_mean', 0):6.0f} MHz  "
          f"{s.get('power_w_mean', 0):6.0f} W  ({s.get('busy_samples')} samples)   {us * 1e-6 * s.get('power_w_mean', 0):.3f} J per launch", flush=True)
net.close()
