"""HF-CLIP reference scorer — TEST INFRASTRUCTURE, not product code.

The reference's arithmetic lives in `transformers.CLIPModel` (third-party; reference call sites
utils/train_eval_util.py:23, utils/detection_util.py:225,229-230).  This module drives that library with
a restatement of the reference's per-batch body (utils/detection_util.py:223-248) so that tests,
`bench.py`'s `cpu_baseline` / `parity.vs_hf` legs and `__graft_entry__.smoke()` can compare the HIP path with
"the HF-CLIP reference" itself (BASELINE.json north_star) on any torch device:

  * `device="cpu"`  — the reference's own configuration on the host cores (the CPU baseline);
  * `device="cuda"` — the same fp32 eager model on the GPU box's device, which scores the 50 000 + 10 000
    images of the headline configuration in minutes.  torch eager fp32 with `attn_implementation="eager"`;
    gfx950 has no TF32/xf32 path (MI355X_MICROARCH.md), `allow_tf32` is forced off anyway.

Only tests/, bench.py and smoke() may import this file (like everything under oracle/); nothing under
mcm_amd/ does — `mcm_amd.parity.measure_drift` takes the scorer as a plain callable.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def _pooled(x):
    """transformers 5.x returns a ModelOutput from get_*_features, 4.x the tensor."""
    return x.pooler_output if hasattr(x, "pooler_output") else x


class HFReference:
    """`CLIPModel` with the build's seeded (or any HF-named) state dict, fp32, eval, on `device`."""

    def __init__(self, geo, state_dict: Dict[str, np.ndarray], device: str = "cpu"):
        import torch
        from transformers import CLIPModel

        torch.backends.cuda.matmul.allow_tf32 = False
        torch.backends.cudnn.allow_tf32 = False
        cfg = geo.hf_configs()
        try:  # plain matmul → softmax → matmul attention, not a fused SDPA backend
            cfg._attn_implementation = "eager"
            cfg.text_config._attn_implementation = "eager"
            cfg.vision_config._attn_implementation = "eager"
        except Exception:
            pass
        self.geo = geo
        self.device = torch.device(device)
        self.model = CLIPModel(cfg).eval()
        missing, unexpected = self.model.load_state_dict(
            {k: torch.from_numpy(np.array(v, dtype=np.float32)) for k, v in state_dict.items()},   # (a copy: the seeded arrays are read-only)
            strict=False)
        bad = [m for m in missing if not m.endswith("position_ids") and m != "logit_scale"]  # unused by MCM
        if bad or unexpected:
            raise RuntimeError(f"HFReference: state dict mismatch (missing {bad[:4]}, unexpected {list(unexpected)[:4]})")
        self.model = self.model.float().to(self.device)
        self._bank = None

    # -- the two calls of the model contract (reference utils/detection_util.py:225,229-230) -----
    def image_features(self, pixel_values, normalize: bool = True):
        import torch

        with torch.no_grad():
            f = _pooled(self.model.get_image_features(pixel_values=pixel_values.to(self.device).float())).float()
            return f / f.norm(dim=-1, keepdim=True) if normalize else f

    def text_features(self, input_ids, attention_mask=None, normalize: bool = True):
        import torch

        ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long, device=self.device)
        mask = None if attention_mask is None else torch.as_tensor(np.asarray(attention_mask), dtype=torch.long,
                                                                   device=self.device)
        with torch.no_grad():
            t = _pooled(self.model.get_text_features(input_ids=ids, attention_mask=mask)).float()
            return t / t.norm(dim=-1, keepdim=True) if normalize else t

    def set_bank(self, input_ids, attention_mask=None):
        """Encode the prompt bank once (the hoisted form; `score_batch(..., reencode=True)` is the reference's)."""
        self._ids, self._mask = input_ids, attention_mask
        self._bank = self.text_features(input_ids, attention_mask)
        return self._bank

    def score_batch(self, pixel_values, T: float = 1.0, score: str = "MCM", reencode: bool = False):
        """One iteration of the reference loop (utils/detection_util.py:223-248) → [b] float32 tensor on
        `device`; `reencode=True` re-runs the text tower like the reference does every batch (:228-231)."""
        import torch

        f = self.image_features(pixel_values)
        t = self.text_features(self._ids, self._mask) if reencode else self._bank
        out = f @ t.T
        if score == "max-logit":
            return -out.max(dim=1).values
        if score == "energy":
            return -(T * torch.logsumexp(out / T, dim=1))
        p = torch.softmax(out / T, dim=1)
        if score == "MCM":
            return -p.max(dim=1).values
        if score == "entropy":  # scipy.stats.entropy(p, axis=1): natural log, rows already sum to 1
            return -(p * torch.log(p.clamp_min(1e-45))).sum(dim=1)
        if score == "var":      # -np.var(p, axis=1), ddof = 0
            return -p.var(dim=1, unbiased=False)
        raise ValueError(score)


def hf_available() -> Optional[str]:
    """None when transformers + CLIPModel import, else the reason."""
    try:
        from transformers import CLIPModel  # noqa: F401

        return None
    except Exception as e:  # pragma: no cover
        return f"{type(e).__name__}: {e}"


def hf_scorer_factory(T: float = 1.0, score: str = "MCM", sub_batch: int = 256):
    """Factory in the form `mcm_amd.parity.measure_drift(external={"hf": ...})` expects: builds the fp32
    HF model on the device the native arms run on, encodes the bank once, and returns `pixels -> scores`."""

    def factory(geo, state_dict, ids, mask, device):
        import torch

        h = HFReference(geo, state_dict, device=str(device))
        h.set_bank(ids, mask)

        def fn(px):
            return torch.cat([h.score_batch(px[s:s + sub_batch], T, score) for s in range(0, px.shape[0], sub_batch)])

        return fn

    return factory
