"""Probe: attention kernel time vs qkv row stride (same FLOPs, same bytes)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from mcm_amd.engine import NativeCLIP
from mcm_amd.config import geometry
from mcm_amd.weights import synth_state_dict
geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="bf16", max_batch=8, max_prompt_tokens=2048)
L = 197
for nseq, heads in ((512, 12), (6144, 1), (1024, 6), (3072, 2)):
    D = heads * 64
    qkv = torch.randn((nseq * L, 3 * D), device="cuda").bfloat16()
    out = torch.empty((nseq * L, D), device="cuda", dtype=torch.bfloat16)
    f = lambda: net._lib.mcm_op_attention(net._h, 0, ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), nseq, L, heads, 0, None)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"nseq={nseq} heads={heads}: {us:.1f} us  ({(qkv.numel()+out.numel())*2/us/1e6:.2f} TB/s)")
