"""Where the `host_raw` ingest leg spends a batch (bench.py ingest_legs' workload: 512 decoded images of 8 sizes, 334 MB):
each stage alone — host pack (mcm_pack_u8) per thread count, the packed H2D copy alone and beside scoring, the device
Resize + CenterCrop, the scoring call — then the pipeline per (depth, threads).   python tools/ingest_probe.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcm_amd.engine import build_model  # noqa: E402
from mcm_amd.ingest import PackedImagePipe  # noqa: E402
from mcm_amd.synth import make_token_ids  # noqa: E402

B, K = 512, 1000
net = build_model("ViT-B/16", precision="fp16", max_batch=B, max_prompt_tokens=K * 77)
ids, mask = make_token_ids(K, seed=2)
txt = net.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), normalize=True)
rng = np.random.default_rng(11)
sizes = [(375, 500), (500, 375), (333, 500), (500, 333), (480, 640), (400, 400), (256, 341), (600, 800)]
base = {hw: rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8) for hw in sizes}
batch = [base[sizes[i % len(sizes)]] for i in range(B)]
nbytes = PackedImagePipe.packed_bytes(batch)
sc = torch.empty(B, device=net.device)


def out(**kw):
    print(json.dumps(kw), flush=True)


out(stage="workload", images=B, MB=round(nbytes / 1e6, 1), host_cpus=os.cpu_count())
for th in (1, 4, 8, 16, 32, 64):
    pipe = PackedImagePipe(net, B, nbytes + (1 << 20), pack_threads=th)
    s = pipe.stage(0)
    pipe._fill(s, batch)
    t0 = time.perf_counter()
    for _ in range(5):
        pipe._fill(s, batch)
    dt = (time.perf_counter() - t0) / 5
    out(stage="pack", threads=th, ms=round(dt * 1e3, 2), GBps=round(nbytes / dt / 1e9, 1))
    del pipe
pipe = PackedImagePipe(net, B, nbytes + (1 << 20), pack_threads=16)
s = pipe.stage(0)
_, offs, hs, ws, nb = pipe._fill(s, batch)


def ev_ms(fn, reps, stream=None):
    st = stream or torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = ev_ms(lambda: s.dev[:nb].copy_(s.host[:nb], non_blocking=True), 5)
out(stage="h2d_copy_alone", ms=round(ms, 2), GBps=round(nb / ms / 1e6, 1))
crops = torch.empty((B, 224, 224, 3), dtype=torch.uint8, device=net.device)
ms = ev_ms(lambda: net.resize_crop_packed(s.dev, offs, hs, ws, out=crops), 10)
out(stage="resize_crop", ms=round(ms, 3))
ms_score = ev_ms(lambda: net.score_images(crops, txt, 1.0, "MCM", out=sc), 10)
out(stage="score_u8", ms=round(ms_score, 3))
# the copy beside scoring: copies back to back on the copy stream while the compute stream scores
cs = torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
with torch.cuda.stream(cs):
    e0.record()
    for _ in range(6):
        s.dev[:nb].copy_(s.host[:nb], non_blocking=True)
    e1.record()
t0 = time.perf_counter()
n = 0
while not e1.query():
    net.score_images(crops, txt, 1.0, "MCM", out=sc)
    n += 1
    if n % 2 == 0:
        torch.cuda.current_stream().synchronize()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out(stage="h2d_copy_beside_scoring", copy_ms=round(e0.elapsed_time(e1) / 6, 2), GBps=round(nb * 6 / e0.elapsed_time(e1) / 1e6, 1),
    score_ms_meanwhile=round(dt * 1e3 / n, 2))
del pipe


class Ablated(PackedImagePipe):
    """the pipeline with the host pack and / or the H2D copy left out (stale bytes: timing only)"""

    def __init__(self, *a, pack=True, copy=True, **kw):
        super().__init__(*a, **kw)
        self.do_pack, self.do_copy, self._memo = pack, copy, {}

    def _fill(self, s, images):
        if self.do_pack or id(s) not in self._memo:
            self._memo[id(s)] = super()._fill(s, images)
        return self._memo[id(s)]

    def push(self, s, nbytes):
        if self.do_copy or not s.used:
            super().push(s, nbytes)


for pack, copy in ((True, True), (False, True), (True, False), (False, False)):
    pipe = Ablated(net, B, nbytes + (1 << 20), depth=3, pack_threads=8, pack=pack, copy=copy)
    for px in pipe.stream([batch] * 4):
        net.score_images(px, txt, 1.0, "MCM", out=sc)
    torch.cuda.synchronize()
    steps = 16
    t0 = time.perf_counter()
    for px in pipe.stream(batch for _ in range(steps)):
        net.score_images(px, txt, 1.0, "MCM", out=sc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out(stage="pipeline_ablation", pack=pack, copy=copy, ms_per_batch=round(dt * 1e3 / steps, 2), images_per_s=round(steps * B / dt))
    del pipe
for depth in (3,):
    for th in (8,):
        pipe = PackedImagePipe(net, B, nbytes + (1 << 20), depth=depth, pack_threads=th)
        for px in pipe.stream([batch, batch]):
            net.score_images(px, txt, 1.0, "MCM", out=sc)
        torch.cuda.synchronize()
        steps = 12
        t0 = time.perf_counter()
        for px in pipe.stream(batch for _ in range(steps)):
            net.score_images(px, txt, 1.0, "MCM", out=sc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out(stage="pipeline", depth=depth, threads=th, ms_per_batch=round(dt * 1e3 / steps, 2), images_per_s=round(steps * B / dt))
        del pipe
net.close()
