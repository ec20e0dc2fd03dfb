"""HF CLIPModel fp32 on the GPU box's device vs the native arms, same device-generated images
(tools probe for the parity.vs_hf leg): python tests/probes/hf_gpu_probe.py [ckpt n_id n_ood batch weights arms K]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mcm_amd.parity import measure_drift  # noqa: E402
from oracle.hf_reference import hf_scorer_factory  # noqa: E402

a = sys.argv[1:]
ckpt = a[0] if a else "ViT-B/16"
n_id, n_ood = (int(a[1]), int(a[2])) if len(a) > 2 else (2048, 1024)
batch = int(a[3]) if len(a) > 3 else 512
weights = a[4] if len(a) > 4 else "fp16-exact"
arms = tuple(a[5].split(",")) if len(a) > 5 else ("fp16",)
K = int(a[6]) if len(a) > 6 else 1000
t0 = time.time()
d = measure_drift(ckpt, K=K, n_id=n_id, n_ood=n_ood, batch=batch, arms=arms, amp=1.5, tile=0.0, weights=weights,
                  external={"hf": hf_scorer_factory()})
d["seconds"] = time.time() - t0
print(json.dumps(d))
