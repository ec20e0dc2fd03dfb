"""Host entropy decoder alone (mcm_jpeg_entropy_decode) per thread count, into pinned and into pageable memory.
    python tools/jpeg_entropy_probe.py"""
import ctypes
import io
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image  # noqa: E402

from mcm_amd.config import JpegImage  # noqa: E402
from mcm_amd.engine import load_library  # noqa: E402

lib = load_library()
rng = np.random.default_rng(13)
sizes = [(375, 500), (500, 375), (333, 500), (500, 333), (480, 640), (400, 400), (256, 341), (600, 800)]
yy, xx = np.mgrid[0:800, 0:800].astype(np.float32)
root = tempfile.mkdtemp(prefix="mcm_jp_")
try:
    paths = []
    for i in range(16):
        h, w = sizes[i % 8]
        f = rng.uniform(0.01, 0.06, 6)
        im = np.stack([127 + 70 * np.sin(f[2 * c] * xx[:h, :w] + i) * np.cos(f[2 * c + 1] * yy[:h, :w]) for c in range(3)], -1)
        im = np.clip(im + rng.normal(0, 12, im.shape), 0, 255).astype(np.uint8)
        p = os.path.join(root, f"{i}.jpg")
        Image.fromarray(im).save(p, quality=90)
        paths.append(p)
    n = 512
    files = [paths[i % 16] for i in range(n)]
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in files])
    meta = (JpegImage * n)()
    quant = np.zeros((n, 3, 64), dtype=np.uint16)
    used = ctypes.c_int64(0)
    lib.mcm_jpeg_entropy_decode(arr, n, None, 0, meta, quant.ctypes.data, 4, ctypes.byref(used))
    import torch

    bufs = {"pageable": torch.empty(used.value, dtype=torch.uint8), "pinned": torch.empty(used.value, dtype=torch.uint8, pin_memory=True)}
    t0 = time.perf_counter()
    for p in files[:64]:
        np.asarray(Image.open(p).convert("RGB"))
    print(json.dumps({"pillow_one_thread_images_per_s": round(64 / (time.perf_counter() - t0)), "coef_MB_per_batch": round(used.value / 1e6, 1)}), flush=True)
    for kind, buf in bufs.items():
        for th in (1, 4, 8, 12, 15, 16, 24, 32):
            lib.mcm_jpeg_entropy_decode(arr, n, buf.data_ptr(), buf.numel(), meta, quant.ctypes.data, th, ctypes.byref(used))
            t0 = time.perf_counter()
            for _ in range(3):
                rc = lib.mcm_jpeg_entropy_decode(arr, n, buf.data_ptr(), buf.numel(), meta, quant.ctypes.data, th, ctypes.byref(used))
            dt = (time.perf_counter() - t0) / 3
            print(json.dumps({"memory": kind, "threads": th, "rc": rc, "ms_per_batch": round(dt * 1e3, 2), "images_per_s": round(n / dt),
                              "per_image_thread_ms": round(dt * 1e3 * min(th, 16) / n, 3)}), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
