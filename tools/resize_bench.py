"""Time mcm_resize_crop_u8 alone: B images of one size class per launch, device-resident, HIP events.
    python tools/resize_bench.py [--batch 512] [--reps 20]
Prints one JSON line per size class: ms per launch, images/s, source GB/s."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcm_amd.config import TEST_GEOMETRIES  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--fused-only", action="store_true", help="harness library, the rounds-2/3 form for every workgroup")
a = ap.parse_args()
geo = TEST_GEOMETRIES["B16-2L"]
net = NativeCLIP(geo, synth_state_dict(geo, seed=0), precision="fp16", max_batch=a.batch, max_prompt_tokens=64 * 16,
                 harness=a.fused_only)
if a.fused_only:
    assert net._lib.mcm_debug_resize_fused_only(1) == 0
rng = np.random.default_rng(0)
for h, w in ((375, 500), (500, 375), (333, 500), (480, 640), (600, 800), (768, 1024), (1200, 1600), (256, 256), (224, 224), (160, 120)):
    one = h * w * 3
    packed = torch.from_numpy(rng.integers(0, 256, a.batch * one, dtype=np.uint8)).cuda()
    offs = [i * one for i in range(a.batch)]
    out = torch.empty((a.batch, 224, 224, 3), device="cuda", dtype=torch.uint8)
    for _ in range(3):
        net.resize_crop_packed(packed, offs, [h] * a.batch, [w] * a.batch, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        net.resize_crop_packed(packed, offs, [h] * a.batch, [w] * a.batch, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(json.dumps({"form": "fused" if a.fused_only else "auto", "size": f"{h}x{w}", "batch": a.batch, "ms": round(ms, 4), "images_per_s": round(a.batch / ms * 1e3),
                      "source_GBps": round(a.batch * one / ms / 1e6, 1)}), flush=True)
net.close()
