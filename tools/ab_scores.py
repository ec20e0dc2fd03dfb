"""A/B helper: scores of one seeded batch through a given build of the library -> .npy (compare two builds bit for bit):
python tools/ab_scores.py mcm_amd/libmcm_hip_atomic.so /tmp/a.npy [batch] [ckpt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mcm_amd.engine as eng  # noqa: E402

eng.LIB_PATH = os.path.abspath(sys.argv[1])
from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.synth import make_token_ids  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
geo = geometry(sys.argv[4] if len(sys.argv) > 4 else "ViT-B/16")
net = eng.NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=B, max_prompt_tokens=16000)
ids, _ = make_token_ids(1000, seed=2)
bank = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
px = torch.randn((B, 3, geo.image_size, geo.image_size), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
s = net.score_images(px, bank)
f = net.get_image_features(px)
np.save(sys.argv[2], np.concatenate([s.cpu().numpy(), f.cpu().numpy().reshape(-1)]))
net.close()
