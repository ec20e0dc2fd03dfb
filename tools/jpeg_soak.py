"""Soak of the JPEG device route: the image-folder loader over a 16 384-file tree, pass after pass, with host RSS, device memory
and the rate of every pass — a leak, a stall or a slow-down over time would show here.   python tools/jpeg_soak.py [--passes 12]"""
import argparse
import io
import json
import os
import resource
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402

from mcm_amd.engine import build_model  # noqa: E402
from mcm_amd.folder import ImageFolderU8  # noqa: E402
from mcm_amd.synth import make_token_ids  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=12)
a = ap.parse_args()
B, K = 512, 1000
net = build_model("ViT-B/16", precision="fp16", max_batch=B, max_prompt_tokens=K * 77)
ids, mask = make_token_ids(K, seed=2)
txt = net.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), normalize=True)
rng = np.random.default_rng(5)
sizes = [(375, 500), (500, 375), (333, 500), (500, 333), (480, 640), (400, 400), (256, 341), (600, 800)]
yy, xx = np.mgrid[0:800, 0:800].astype(np.float32)
root = tempfile.mkdtemp(prefix="mcm_soak_")
try:
    for i in range(32):
        h, w = sizes[i % 8]
        f = rng.uniform(0.01, 0.06, 6)
        im = np.stack([127 + 70 * np.sin(f[2 * c] * xx[:h, :w] + i) * np.cos(f[2 * c + 1] * yy[:h, :w]) for c in range(3)], -1)
        im = np.clip(im + rng.normal(0, 12, im.shape), 0, 255).astype(np.uint8)
        Image.fromarray(im).save(os.path.join(root, f"blob{i}.bin"), format="JPEG", quality=90, progressive=(i % 8 == 7))
    for c in range(16):
        os.makedirs(os.path.join(root, f"class{c:02d}"))
    n = 32 * B
    for i in range(n):
        os.link(os.path.join(root, f"blob{i % 32}.bin"), os.path.join(root, f"class{i % 16:02d}", f"{i:06d}.jpg"))
    loader = ImageFolderU8(root, net, B)
    sc = torch.empty(B, device=net.device)
    first = None
    for p in range(a.passes):
        t0 = time.perf_counter()
        acc = []
        for px, _ in loader:
            net.score_images(px, txt, 1.0, "MCM", out=sc[: px.shape[0]])
            acc.append(sc[: px.shape[0]].clone())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        scores = torch.cat(acc)
        if first is None:
            first = scores
        print(json.dumps({"pass": p, "images_per_s": round(n / dt), "host_rss_MB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024,
                          "device_MB": torch.cuda.memory_allocated() // (1 << 20), "same_scores_as_pass_0": bool(torch.equal(scores, first))}), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
    net.close()
