"""Throughput of mcm_resize_crop_u8 on ImageNet-sized inputs (loader-side work).
Usage: python tools/preprocess_probe.py [H W [batch]]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from mcm_amd.config import TEST_GEOMETRIES  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

H, W, B = (int(v) for v in (sys.argv[1:4] + ["375", "500", "512"][len(sys.argv) - 1:]))
geo = TEST_GEOMETRIES["B16-2L"]
net = NativeCLIP(geo, synth_state_dict(geo, 0), max_batch=B, max_prompt_tokens=1024)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(B)]
for _ in range(3):
    out = net.resize_crop(imgs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
n = 20
for _ in range(n):
    out = net.resize_crop(imgs)
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
dev = e0.elapsed_time(e1) * 1e-3 / n
byts = B * (H * W * 3 + 224 * 224 * 3)
print(f"{H}x{W} x{B}: wall {B / wall:,.0f} img/s, device stream {B / dev:,.0f} img/s, "
      f"{byts / dev / 1e9:.0f} GB/s algorithmic (source read + crop written)")
