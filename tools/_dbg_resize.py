import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcm_amd.config import TEST_GEOMETRIES
from mcm_amd.engine import NativeCLIP
from mcm_amd.weights import synth_state_dict
from oracle import oracle as orc
geo = TEST_GEOMETRIES["B16-2L"]
net = NativeCLIP(geo, synth_state_dict(geo, seed=0), precision="fp16", max_batch=32, max_prompt_tokens=64 * 16)
def run(img, tag):
    got = net.resize_crop([torch.from_numpy(img).cuda()]).cpu().numpy()[0]
    ref = orc.resize_crop_u8(img, 224)
    bad = got != ref
    flat_bad = bad.reshape(224, 672)
    print(tag, img.shape, "bad", round(bad.mean(), 4), "by out byte lane", [round(flat_bad[:, l::4].mean(), 3) for l in range(4)],
          "by row%8", [round(bad[r::8].mean(), 3) for r in range(8)])
    ys, xs, cs = np.nonzero(bad)
    for y, x, c in list(zip(ys, xs, cs))[:5]:
        print("   ", y, x, c, "got", got[y, x, c], "ref", ref[y, x, c])
for h, w in ((224, 224), (224, 300), (300, 224), (375, 500)):
    run(np.full((h, w, 3), 100, dtype=np.uint8), "const100")
    ramp = np.zeros((h, w, 3), dtype=np.uint8); ramp[:] = (np.arange(w) % 256)[None, :, None]
    run(ramp, "xramp")
    ramp = np.zeros((h, w, 3), dtype=np.uint8); ramp[:] = (np.arange(h) % 256)[:, None, None]
    run(ramp, "yramp")
    run(np.random.default_rng(h * 10007 + w).integers(0, 256, (h, w, 3), dtype=np.uint8), "random")
