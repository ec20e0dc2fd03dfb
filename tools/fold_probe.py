"""Time the plain vs folded-LayerNorm GEMM epilogues at the model's shapes through the C ABI."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from mcm_amd.config import geometry
from mcm_amd.engine import NativeCLIP
from mcm_amd.weights import synth_state_dict
geo = geometry("tiny")
net = NativeCLIP(geo, synth_state_dict(geo, 0), precision="fp16", max_batch=8, max_prompt_tokens=2048)
lib, h = net._lib, net._h
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
M = 100864
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
g = torch.Generator(device="cuda").manual_seed(0)
for (N, K, gelu) in ((2304, 768, 0), (3072, 768, 1)):
    x = torch.randn((M, K), device="cuda", generator=g).half()
    w = (torch.randn((N, K), device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(N, device="cuda", generator=g)
    cs = torch.randn(N, device="cuda", generator=g)
    rowab = torch.rand((M, 2), device="cuda", generator=g) + 0.5
    y = torch.empty((M, N), device="cuda", dtype=torch.float16)
    t0 = timeit(lambda: lib.mcm_op_linear(h, 2, P(x), P(w), P(b), P(y), None, M, N, K, gelu, None))
    t1 = timeit(lambda: lib.mcm_op_linear_folded(h, 2, P(x), P(rowab), P(w), P(cs), P(b), P(y), M, N, K, gelu, None))
    print(f"consumer N={N} K={K} gelu={gelu}: plain {t0:.1f} us, folded {t1:.1f} us", flush=True)
for (N, K) in ((768, 768), (768, 3072)):
    x = torch.randn((M, K), device="cuda", generator=g).half()
    w = (torch.randn((N, K), device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(N, device="cuda", generator=g)
    gam = torch.randn(N, device="cuda", generator=g)
    r = torch.randn((M, N), device="cuda", generator=g)
    xg = torch.empty((M, N), device="cuda", dtype=torch.float16)
    stats = torch.empty((M, N // 64, 2), device="cuda")
    rowab = torch.empty((M, 2), device="cuda")
    t0 = timeit(lambda: lib.mcm_op_linear(h, 2, P(x), P(w), P(b), None, P(r), M, N, K, 2, None))
    t1 = timeit(lambda: lib.mcm_op_resid_ln(h, 2, P(x), P(w), P(b), P(r), P(gam), P(xg), P(stats), P(rowab), M, N, K, 1e-5, None))
    ln = torch.empty((M, N), device="cuda", dtype=torch.float16)
    t2 = timeit(lambda: lib.mcm_op_layernorm(h, 2, P(r), P(gam), P(b), P(ln), M, N, 1e-5, 0, None))
    print(f"producer N={N} K={K}: plain resid {t0:.1f} us, resid+xg+stats+finalize {t1:.1f} us; standalone LayerNorm {t2:.1f} us", flush=True)
net.close()
