"""Where the JPEG ingest leg spends its time (bench.py ingest_legs host_jpeg): the decode pool alone per worker count, then the
whole loader (decode -> pack -> copy -> resize -> score).   python tools/decode_probe.py"""
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

from mcm_amd.folder import ImageFolderU8  # noqa: E402


def out(**kw):
    print(json.dumps(kw), flush=True)


B = 512
t0 = time.perf_counter()
rng = np.random.default_rng(13)
sizes = [(375, 500), (500, 375), (333, 500), (500, 333), (480, 640), (400, 400), (256, 341), (600, 800)]
yy, xx = np.mgrid[0:800, 0:800].astype(np.float32)
blobs = []
for i in range(16):
    h, w = sizes[i % 8]
    f = rng.uniform(0.01, 0.06, 6)
    im = np.stack([127 + 70 * np.sin(f[2 * c] * xx[:h, :w] + i) * np.cos(f[2 * c + 1] * yy[:h, :w]) for c in range(3)], -1)
    im = np.clip(im + rng.normal(0, 12, im.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(im).save(buf, format="JPEG", quality=90)
    blobs.append(buf.getvalue())
root = tempfile.mkdtemp(prefix="mcm_jpeg_")
n = 8 * B
for i in range(n):
    d = os.path.join(root, f"class{i % 8}")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"{i:05d}.jpg"), "wb") as fh:
        fh.write(blobs[i % 16])
out(stage="generate", seconds=round(time.perf_counter() - t0, 2), files=n, host_cpus=os.cpu_count())
try:
    t0 = time.perf_counter()
    a = [np.asarray(Image.open(io.BytesIO(blobs[i % 16])).convert("RGB")) for i in range(64)]
    out(stage="decode_one_core", images_per_s=round(64 / (time.perf_counter() - t0)))
    for nw in (8, 16, 32, 64, 128):
        ld = ImageFolderU8(root, None, B, workers=nw)
        t0 = time.perf_counter()
        it = ld.decoded_batches(copy=False)
        first = next(it)
        t1 = time.perf_counter()
        m = len(first[0])
        for imgs, _ in it:
            m += len(imgs)
        t2 = time.perf_counter()
        m2 = sum(len(imgs) for imgs, _ in ld.decoded_batches(copy=False))   # second pass: every page of the slots exists
        t3 = time.perf_counter()
        ld.close()
        out(stage="pool_only", workers=nw, start_and_first_batch_s=round(t1 - t0, 2), images_per_s=round((m - B) / (t2 - t1)),
            second_pass_images_per_s=round(m2 / (t3 - t2)), close_s=round(time.perf_counter() - t3, 2))
    import torch

    from mcm_amd.engine import build_model
    from mcm_amd.synth import make_token_ids

    K = 1000
    net = build_model("ViT-B/16", precision="fp16", max_batch=B, max_prompt_tokens=K * 77)
    ids, mask = make_token_ids(K, seed=2)
    txt = net.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), normalize=True)
    sc = torch.empty(B, device=net.device)
    for nw in (16, 32, 64):
        ld = ImageFolderU8(root, net, B, workers=nw)
        t0 = time.perf_counter()
        for px, _ in ld:
            net.score_images(px, txt, 1.0, "MCM", out=sc[: px.shape[0]])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for px, _ in ld:
            net.score_images(px, txt, 1.0, "MCM", out=sc[: px.shape[0]])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ld.close()
        out(stage="loader", workers=nw, first_pass_s=round(t1 - t0, 2), second_pass_images_per_s=round(n / (t2 - t1)))
    net.close()
finally:
    shutil.rmtree(root, ignore_errors=True)
