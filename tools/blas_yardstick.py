"""Known-good yardstick (guide §5.4 rule 10): what the vendor GEMM (hipBLASLt / rocBLAS through torch.matmul)
reaches on the four vision GEMM shapes of B/16 at batch 512, uniform random data, no fused epilogue.
Not used by the product — a ceiling estimate for the hand-written kernels."""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100864
shapes = {"qkv": (2304, 768), "out": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
for dt in (torch.bfloat16, torch.float16):
    for name, (N, K) in shapes.items():
        x = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
        w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(dt)
        b = torch.rand(N, device="cuda").to(dt)
        for _ in range(3):
            y = torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(10):
                y = torch.nn.functional.linear(x, w, b)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"{str(dt):16s} {name:4s} M={M} N={N} K={K}: {best*1e3:8.1f} us  {2.0*M*N*K/best/1e9:8.1f} TFLOP/s", flush=True)
