"""Two half-batch scoring pipelines on two streams sharing the chip (EXPERIMENTS.md R4.11): do the HBM-bound phases of one
(LayerNorm, attention, the residual GEMMs' epilogues) overlap the MFMA-bound phases of the other?  The persistent GEMMs are
given half the CUs each (harness switch mcm_debug_persistent_grid) so that both can be resident.
    python tools/dual_stream_probe.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcm_amd.engine import build_model  # noqa: E402
from mcm_amd.synth import make_token_ids  # noqa: E402

K, B, STEPS = 1000, 512, 24
ids, mask = make_token_ids(K, seed=2)


def out(**kw):
    print(json.dumps(kw), flush=True)


def run(nets, batch, grid, stagger_ms, steps=STEPS):
    lib = nets[0]._lib
    assert lib.mcm_debug_persistent_grid(grid) == 0
    streams = [torch.cuda.Stream() for _ in nets]
    gen = torch.Generator(device="cuda").manual_seed(5)
    px = [torch.randn((batch, 3, 224, 224), generator=gen, device="cuda") for _ in nets]
    sc = [torch.empty(batch, device="cuda") for _ in nets]

    def step(i):
        with torch.cuda.stream(streams[i]):
            nets[i].score_images(px[i], txts[i], 1.0, "MCM", out=sc[i])

    for i in range(len(nets)):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(0)
    if len(nets) > 1 and stagger_ms > 0:
        time.sleep(stagger_ms * 1e-3)
    for s in range(steps):
        for i in range(len(nets)):
            if s == 0 and i == 0:
                continue
            step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.mcm_debug_persistent_grid(0)
    return len(nets) * steps * batch / dt


nets = [build_model("ViT-B/16", precision="fp16", max_batch=B, max_prompt_tokens=K * 77, harness=True) for _ in range(2)]
txts = [n.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), normalize=True) for n in nets]
for rep in range(2):
    out(form="one stream, batch 512, one workgroup per CU", images_per_s=round(run(nets[:1], 512, 0, 0)))
    out(form="one stream, batch 256, one workgroup per CU", images_per_s=round(run(nets[:1], 256, 0, 0)))
    out(form="two streams x batch 256, full grids", stagger_ms=0, images_per_s=round(run(nets, 256, 0, 0)))
    for grid in (128, 160, 192):
        for stagger in (0, 2, 5):
            out(form="two streams x batch 256", grid=grid, stagger_ms=stagger, images_per_s=round(run(nets, 256, grid, stagger)))
    out(form="two streams x batch 512", grid=128, stagger_ms=5, images_per_s=round(run(nets, 512, 128, 5, steps=12)))
for n in nets:
    n.close()
