"""Exactness sweep of the JPEG device route against Pillow: N random JPEG files (sizes 1 ... 700, quality 1 ... 100, 4:4:4 /
4:2:2 / 4:2:0 / grayscale, baseline / progressive, optimised tables, restart markers; noise, photograph-like, flat, saturated
checkerboards, hard edges) -> mcm_jpeg_entropy_decode + mcm_jpeg_reconstruct on the MI355X vs Image.open().convert("RGB").
    python tools/jpeg_sweep.py [--n 1000]"""
import argparse
import ctypes
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402

from mcm_amd.config import JpegImage, geometry  # noqa: E402
from mcm_amd.engine import NativeCLIP  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000)
a = ap.parse_args()
geo = geometry("B16-2L")
B = 50
net = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=B, max_prompt_tokens=1024)
lib = net._lib
rng = np.random.default_rng(2024)
root = tempfile.mkdtemp(prefix="mcm_sweep_")
stats = {"files": 0, "taken": 0, "not_taken": 0, "equal": 0, "differ": [], "save_failed": 0, "by_kind": {}}
try:
    def content(kind, h, w):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        if kind == "noise":
            return rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        if kind == "photo":
            f = rng.uniform(0.005, 0.3, 6)
            im = np.stack([127 + 110 * np.sin(f[2 * c] * xx + c) * np.cos(f[2 * c + 1] * yy) for c in range(3)], -1)
            return np.clip(im + rng.normal(0, rng.uniform(0, 30), im.shape), 0, 255).astype(np.uint8)
        if kind == "flat":
            return np.broadcast_to(rng.integers(0, 256, 3).astype(np.uint8), (h, w, 3)).copy()
        if kind == "checker":
            p = int(rng.integers(1, 9))
            m = (((yy // p) + (xx // p)) % 2).astype(np.uint8) * 255
            return np.stack([m, 255 - m, m], -1)
        e = np.zeros((h, w, 3), np.uint8)
        for _ in range(12):
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            e[y0: y0 + int(rng.integers(1, 40)), x0: x0 + int(rng.integers(1, 60))] = rng.integers(0, 256, 3)
        return e

    done = 0
    while done < a.n:
        paths, kinds = [], []
        while len(paths) < B and done + len(paths) < a.n:
            h = int(rng.integers(1, 700)) if rng.random() < 0.8 else int(rng.integers(1, 20))
            w = int(rng.integers(1, 700)) if rng.random() < 0.8 else int(rng.integers(1, 20))
            kind = ["noise", "photo", "photo", "flat", "checker", "edges"][int(rng.integers(0, 6))]
            q = int(rng.integers(1, 101))
            kw = dict(quality=q)
            gray = rng.random() < 0.1
            if not gray:
                kw["subsampling"] = int(rng.integers(0, 3))
            if rng.random() < 0.3:
                kw["progressive"] = True
            elif rng.random() < 0.5 and q <= 85:
                kw["optimize"] = True
            if rng.random() < 0.1:
                kw["restart_marker_blocks"] = int(rng.integers(1, 40))
            im = content(kind, h, w)
            p = os.path.join(root, f"{done + len(paths):05d}.jpg")
            try:
                Image.fromarray(im[:, :, 0] if gray else im).save(p, **kw)
            except Exception:
                stats["save_failed"] += 1
                continue
            paths.append(p)
            kinds.append((kind, h, w, kw, gray))
        n = len(paths)
        arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
        meta = (JpegImage * B)()
        quant = np.zeros((B, 3, 64), dtype=np.uint16)
        used = ctypes.c_int64(0)
        lib.mcm_jpeg_entropy_decode(arr, n, None, 0, meta, quant.ctypes.data, 8, ctypes.byref(used))
        buf = torch.empty(max(16, used.value), dtype=torch.uint8, pin_memory=True)
        rc = lib.mcm_jpeg_entropy_decode(arr, n, buf.data_ptr(), buf.numel(), meta, quant.ctypes.data, 8, ctypes.byref(used))
        assert rc == 0, rc
        want = []
        for p in paths:
            with Image.open(p) as im:
                want.append(np.asarray(im.convert("RGB")))
        offs, o = [], 0
        for w_ in want:
            offs.append(o)
            o += (w_.size + 15) // 16 * 16
        rgb = torch.zeros(o + 16, dtype=torch.uint8, device="cuda")
        net.jpeg_reconstruct(buf.cuda(), meta, quant, n, rgb, offs)
        got = rgb.cpu().numpy()
        for i in range(n):
            stats["files"] += 1
            key = f"{'progressive' if kinds[i][3].get('progressive') else 'baseline'}/{'gray' if kinds[i][4] else kinds[i][3].get('subsampling')}"
            rec = stats["by_kind"].setdefault(key, [0, 0, 0])
            if meta[i].status != 0:
                stats["not_taken"] += 1
                rec[1] += 1
                continue
            stats["taken"] += 1
            g = got[offs[i]: offs[i] + want[i].size].reshape(want[i].shape)
            if np.array_equal(g, want[i]):
                stats["equal"] += 1
                rec[0] += 1
            else:
                rec[2] += 1
                d = np.abs(g.astype(int) - want[i].astype(int))
                stats["differ"].append({"file": os.path.basename(paths[i]), "case": str(kinds[i]), "max": int(d.max()), "frac": float((d > 0).mean())})
        done += n
    stats["by_kind"] = {k: {"equal": v[0], "not_taken": v[1], "differ": v[2]} for k, v in sorted(stats["by_kind"].items())}
    print(json.dumps(stats))
finally:
    shutil.rmtree(root, ignore_errors=True)
    net.close()
