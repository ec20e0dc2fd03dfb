"""Exactness sweep of mcm_resize_crop_u8 against Pillow itself: N random image sizes (1 ... 1500 per side, photograph-like
and noise content) -> torchvision's Resize(224) + CenterCrop(224) arithmetic with Pillow's Image.resize(BILINEAR) on the host
vs the device kernel.    python tools/resize_sweep.py [--n 600]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402

from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402


def pillow_route(a, S=224):
    """torchvision.transforms.Resize(S) (short side -> S, long side int(S * long / short), untouched when short == S) and
    CenterCrop(S) (top / left = round((n - S) / 2)) around Pillow's resize — functional.py resize / center_crop."""
    h, w = a.shape[:2]
    im = Image.fromarray(a)
    short, long_ = (w, h) if w <= h else (h, w)
    if short != S:
        new_short, new_long = S, int(S * long_ / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        im = im.resize((nw, nh), Image.BILINEAR)
    nw, nh = im.size
    top, left = int(round((nh - S) / 2.0)), int(round((nw - S) / 2.0))
    return np.asarray(im.crop((left, top, left + S, top + S)))


ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=600)
a = ap.parse_args()
geo = geometry("B16-2L")
B = 24
net = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=B, max_prompt_tokens=1024)
rng = np.random.default_rng(77)
stats = {"images": 0, "equal": 0, "differ": [], "refused": 0}
done = 0
while done < a.n:
    imgs = []
    for _ in range(min(B, a.n - done)):
        if rng.random() < 0.15:
            h, w = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        else:
            h, w = int(rng.integers(20, 1500)), int(rng.integers(20, 1500))
        if max(h, w) / min(h, w) > 25 and min(h, w) < 224:  # (a scale factor above the kernel's 31: refused, tested elsewhere)
            h = w
        if rng.random() < 0.5:
            im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        else:
            yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
            f = rng.uniform(0.01, 0.5, 6)
            im = np.stack([127 + 120 * np.sin(f[2 * c] * xx) * np.cos(f[2 * c + 1] * yy) for c in range(3)], -1).clip(0, 255).astype(np.uint8)
        imgs.append(im)
    got = net.resize_crop([torch.from_numpy(i).cuda() for i in imgs]).cpu().numpy()
    for im, g in zip(imgs, got):
        stats["images"] += 1
        want = pillow_route(im)
        if np.array_equal(g, want):
            stats["equal"] += 1
        else:
            d = np.abs(g.astype(int) - want.astype(int))
            stats["differ"].append({"size": list(im.shape[:2]), "max": int(d.max()), "frac": float((d > 0).mean())})
    done += len(imgs)
print(json.dumps(stats))
net.close()
