"""BASELINE config 3 end to end from JPEG FILES through the reference's CLI (eval_ood_detection.py --in_dataset ImageNet):
an ImageNet-shaped tree of image folders under --root-dir (50 000 validation images in 1 000 classes + the four OOD sets,
10 000 / 10 000 / 10 000 / 5 640) -> Pillow decode in the loader's worker processes -> packed upload -> Resize + CenterCrop +
scoring + AUROC / AUPR / FPR95 on the device.  The files are hard links to 32 synthetic JPEGs (photograph-like, 8 sizes,
quality 90): what is exercised is the whole real-data route at the real size, not the content.
    python tools/e2e_jpeg_config3.py [--scale 1.0] [--dtype fp16]"""
import argparse
import io
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0, help="fraction of the sets' sizes")
ap.add_argument("--dtype", default="fp16")
a = ap.parse_args()

t0 = time.perf_counter()
root = tempfile.mkdtemp(prefix="mcm_e2e_")
work = tempfile.mkdtemp(prefix="mcm_e2e_run_")
try:
    rng = np.random.default_rng(21)
    sizes = [(375, 500), (500, 375), (333, 500), (500, 333), (480, 640), (400, 400), (256, 341), (600, 800)]
    yy, xx = np.mgrid[0:800, 0:800].astype(np.float32)
    base, nbytes = [], 0
    os.makedirs(os.path.join(root, "_blobs"))
    for i in range(32):
        h, w = sizes[i % 8]
        f = rng.uniform(0.01, 0.06, 6)
        im = np.stack([127 + 70 * np.sin(f[2 * c] * xx[:h, :w] + i) * np.cos(f[2 * c + 1] * yy[:h, :w]) for c in range(3)], -1)
        im = np.clip(im + rng.normal(0, 12, im.shape), 0, 255).astype(np.uint8)
        p = os.path.join(root, "_blobs", f"{i}.jpg")
        Image.fromarray(im).save(p, quality=90)
        base.append(p)
        nbytes += os.path.getsize(p)
    sets = {("ImageNet", "val"): (int(50000 * a.scale), 1000), ("ImageNet_OOD_dataset", "iNaturalist"): (int(10000 * a.scale), 10),
            ("ImageNet_OOD_dataset", "SUN"): (int(10000 * a.scale), 10), ("ImageNet_OOD_dataset", "Places"): (int(10000 * a.scale), 10),
            ("ImageNet_OOD_dataset", "dtd", "images"): (int(5640 * a.scale), 47)}
    total = 0
    for sub, (n, ncls) in sets.items():
        for c in range(ncls):
            os.makedirs(os.path.join(root, *sub, f"n{c:08d}"))
        for i in range(n):
            os.link(base[(i * 7 + len(sub)) % 32], os.path.join(root, *sub, f"n{i % ncls:08d}", f"{i:06d}.jpg"))
        total += n
    t_tree = time.perf_counter() - t0
    import eval_ood_detection as cli

    os.chdir(work)
    t1 = time.perf_counter()
    r = cli.main(["--in_dataset", "ImageNet", "--CLIP_ckpt", "ViT-B/16", "-b", "512", "--dtype", a.dtype, "--name", "e2e",
                  "--root-dir", root, "--score", "MCM"])
    t_cli = time.perf_counter() - t1
    out = {"images": total, "mean_jpeg_bytes": nbytes / 32, "tree_seconds": round(t_tree, 1), "cli_seconds": round(t_cli, 1),
           "images_per_sec_whole_cli": round(total / t_cli), "dtype": a.dtype,
           "sources": {k: (v["kind"], v.get("n")) for k, v in r["sources"].items()},
           "measures_auroc_aupr_fpr": {k: [round(float(x), 6) for x in v] for k, v in r["measures"].items()},
           "refine_rescored": (r.get("refine") or {}).get("rescored"), "host_cpus": os.cpu_count()}
    print(json.dumps(out, default=str))
finally:
    shutil.rmtree(root, ignore_errors=True)
    shutil.rmtree(work, ignore_errors=True)
