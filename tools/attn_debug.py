import ctypes, sys, torch
sys.path.insert(0, ".")
from mcm_amd.config import geometry
from mcm_amd.engine import NativeCLIP
from mcm_amd.weights import synth_state_dict
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
import itertools
for variant, (L, causal), (pr, dtp) in itertools.product((1,), ((16, 1), (40, 1), (64, 0), (50, 0), (257, 0), (128, 1), (65, 0), (80, 0), (96, 0), (112, 0), (197, 0), (77, 1)), ((2, torch.float16), (0, torch.bfloat16))):
    lib.mcm_debug_attention_variant(variant)
    nseq, heads = 2, 2
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(L)
    qkv = torch.randn((nseq * L, 3 * D), device="cuda", generator=g).to(dtp)
    out = torch.zeros((nseq * L, D), device="cuda", dtype=dtp)
    rc = lib.mcm_op_attention(net._h, pr, ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), nseq, L, heads, causal, None)
    torch.cuda.synchronize()
    if rc:
        print("variant", variant, "L", L, "rc", rc)
        continue
    err = (out.double() - ref(qkv, nseq, L, heads, causal)).abs().view(nseq, L, heads, 4, 16)
    bad_q = (err.amax(dim=(0, 2, 3, 4)) > 0.02).nonzero().flatten().tolist()
    bad_dt = (err.amax(dim=(0, 1, 2, 4)) > 0.02).nonzero().flatten().tolist()
    print(f"variant={variant} prec={pr} L={L} causal={causal}: max err {err.max().item():.3e}; bad queries {bad_q[:40]}{'...' if len(bad_q) > 40 else ''} ({len(bad_q)}); bad dt {bad_dt}", flush=True)
net.close()
