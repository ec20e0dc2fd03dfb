"""Parameter schema (HF state_dict names) and the build-owned seeded
weight generator.

No CLIP checkpoint exists in either container (hub offline), so parity is proven on
seeded synthetic parameters that both containers regenerate bit-identically: every
tensor is drawn from its own counter-based Philox stream keyed by (seed, crc32(name)),
so the values do not depend on generation order.  A real checkpoint, when present, is
loaded through the same names (`load_state_dict_file`).

Schema source: transformers modeling_clip.py:138-159, 221-230, 280-296, 338-344, 353-360,
494-508, 594-608, 661-679.
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np

from .config import ClipGeometry


def _layer_shapes(prefix: str, d: int, ff: int) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
        s[f"{prefix}.self_attn.{proj}.weight"] = (d, d)
        s[f"{prefix}.self_attn.{proj}.bias"] = (d,)
    for ln in ("layer_norm1", "layer_norm2"):
        s[f"{prefix}.{ln}.weight"] = (d,)
        s[f"{prefix}.{ln}.bias"] = (d,)
    s[f"{prefix}.mlp.fc1.weight"] = (ff, d)
    s[f"{prefix}.mlp.fc1.bias"] = (ff,)
    s[f"{prefix}.mlp.fc2.weight"] = (d, ff)
    s[f"{prefix}.mlp.fc2.bias"] = (d,)
    return s


def param_shapes(geo: ClipGeometry) -> Dict[str, Tuple[int, ...]]:
    """Every parameter the path reads, by HF name → shape.  `logit_scale` is omitted: MCM
    never multiplies by it (reference utils/detection_util.py:232)."""
    s: Dict[str, Tuple[int, ...]] = {}
    dv, dt, p = geo.v_width, geo.t_width, geo.patch_size
    s["vision_model.embeddings.class_embedding"] = (dv,)
    s["vision_model.embeddings.patch_embedding.weight"] = (dv, 3, p, p)
    s["vision_model.embeddings.position_embedding.weight"] = (geo.v_tokens, dv)
    s["vision_model.pre_layrnorm.weight"] = (dv,)
    s["vision_model.pre_layrnorm.bias"] = (dv,)
    for i in range(geo.v_layers):
        s.update(_layer_shapes(f"vision_model.encoder.layers.{i}", dv, geo.v_mlp))
    s["vision_model.post_layernorm.weight"] = (dv,)
    s["vision_model.post_layernorm.bias"] = (dv,)
    s["visual_projection.weight"] = (geo.proj_dim, dv)
    s["text_model.embeddings.token_embedding.weight"] = (geo.vocab_size, dt)
    s["text_model.embeddings.position_embedding.weight"] = (geo.max_positions, dt)
    for i in range(geo.t_layers):
        s.update(_layer_shapes(f"text_model.encoder.layers.{i}", dt, geo.t_mlp))
    s["text_model.final_layer_norm.weight"] = (dt,)
    s["text_model.final_layer_norm.bias"] = (dt,)
    s["text_projection.weight"] = (geo.proj_dim, dt)
    return s


def _std_for(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(mean, std) of the synthetic init.  Chosen so every code path carries signal:
    attention logits have O(1) spread (non-uniform softmax), QuickGELU sees O(1)
    pre-activations, biases and LayerNorm affine terms are non-trivial."""
    if name.endswith("norm.weight") or name.endswith("norm1.weight") or \
            name.endswith("norm2.weight") or name.endswith("layrnorm.weight"):
        return 1.0, 0.1
    if name.endswith(".bias"):
        return 0.0, 0.05
    if name.endswith("class_embedding"):
        return 0.0, 0.5
    if "position_embedding" in name:
        return 0.0, 0.1
    if "token_embedding" in name:
        return 0.0, 0.5
    if "patch_embedding" in name:
        fan_in = int(np.prod(shape[1:]))
        return 0.0, fan_in ** -0.5
    fan_in = shape[-1]
    if "q_proj" in name or "k_proj" in name or "fc1" in name:
        return 0.0, fan_in ** -0.5
    if "v_proj" in name or "out_proj" in name or "fc2" in name:
        return 0.0, 0.5 * fan_in ** -0.5
    return 0.0, fan_in ** -0.5  # projections


def synth_param(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    key = (int(seed) << 32) | zlib.crc32(name.encode())
    rng = np.random.Generator(np.random.Philox(key=key))
    mean, std = _std_for(name, shape)
    a = rng.standard_normal(size=shape, dtype=np.float32)
    a *= np.float32(std)
    if mean:
        a += np.float32(mean)
    return np.ascontiguousarray(a)


def synth_state_dict(geo: ClipGeometry, seed: int = 0, regime: str = "fp32") -> Dict[str, np.ndarray]:
    """regime="fp32": the seeded values as drawn (fp32-valued: nothing like an fp16 number — a 16-bit arm then
    runs the split-weight GEMMs, include/mcm.h MCM_WEIGHTS_*).  regime="fp16-exact": every parameter rounded to the
    nearest fp16 value, returned as fp32 — the situation of the reference's checkpoints (`openai/clip-vit-*`: Linear /
    conv / projection weights trained and released in fp16, widened to fp32 by the HF conversion), for which one
    fp16 operand per weight is lossless."""
    if regime not in ("fp16-exact", "fp32"):
        raise ValueError(f"unknown weight regime {regime!r}")
    key = (tuple(sorted(param_shapes(geo).items())), int(seed), regime)
    hit = _SD_CACHE.get(key)
    if hit is None:
        sd = {n: synth_param(n, s, seed) for n, s in param_shapes(geo).items()}
        if regime == "fp16-exact":
            sd = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
        for v in sd.values():
            v.setflags(write=False)   # shared between callers from here on: replace an entry, never write into one
        _SD_CACHE[key] = hit = sd
        while len(_SD_CACHE) > _SD_CACHE_MAX:   # (B/16: 0.6 GB, L/14: 1.7 GB of host memory per entry)
            _SD_CACHE.pop(next(iter(_SD_CACHE)))
    else:
        _SD_CACHE[key] = _SD_CACHE.pop(key)     # most recently used last
    return dict(hit)


# synth_state_dict is a pure function of (shapes, seed, regime) that costs 5 s at ViT-B/16 and 17 s at ViT-L/14 (150 M / 430 M
# counter-based draws); the CLI, bench.py and the tests ask for the same few checkpoints again and again.  The arrays are
# handed out read-only and shared; the dict is the caller's own.
_SD_CACHE: Dict = {}
_SD_CACHE_MAX = 6


def inject_outlier_channels(sd: Dict[str, np.ndarray], geo: ClipGeometry, *, channels: int = 6, scale: float = 100.0,
                            gamma_scale: float = None, seed: int = 7) -> Tuple[Dict[str, np.ndarray], np.ndarray]:
    """Outlier-channel stress checkpoint.  Real CLIP ViTs carry a handful of residual-stream
    channels whose activations are 50 - 200 x the rest ("massive activations"), written by a few rows of `out_proj` /
    `fc2` and read through LayerNorm gains; seeded weights at HF init scales have none, so the fp16 arm's range and
    precision are never exercised where a real checkpoint exercises them.  This returns a copy of `sd` in which
    `channels` residual channels of the VISION tower are such outliers: in every layer the rows of
    `out_proj.weight` / `fc2.weight` (and their bias entries) that write those channels are multiplied by `scale`, and the
    `layer_norm1/2.weight` entries that read them by `gamma_scale` (default: `scale` — the harshest reading of the
    stress: the outlier also reaches the next GEMM's operand amplified).  Returns (state dict, channel indices)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    ch = np.sort(rng.choice(geo.v_width, size=channels, replace=False))
    g = scale if gamma_scale is None else gamma_scale
    out = {k: v.copy() for k, v in sd.items()}
    for i in range(geo.v_layers):
        pre = f"vision_model.encoder.layers.{i}"
        for lin in ("self_attn.out_proj", "mlp.fc2"):
            out[f"{pre}.{lin}.weight"][ch, :] *= np.float32(scale)
            out[f"{pre}.{lin}.bias"][ch] *= np.float32(scale)
        for ln in ("layer_norm1", "layer_norm2"):
            out[f"{pre}.{ln}.weight"][ch] *= np.float32(g)
    return out, ch


def load_state_dict_file(path: str, geo: ClipGeometry) -> Dict[str, np.ndarray]:
    """Read a real checkpoint (`.safetensors` or torch state_dict) by HF names
    (replaces CLIPModel.from_pretrained, reference utils/train_eval_util.py:23)."""
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file

        raw = load_file(path)
    else:
        import torch

        raw = {k: v.float().numpy() for k, v in torch.load(path, map_location="cpu").items()}
    out = {}
    for name, shape in param_shapes(geo).items():
        if name not in raw:
            raise KeyError(f"checkpoint {path} lacks parameter {name}")
        a = np.ascontiguousarray(raw[name], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {a.shape} != expected {shape}")
        out[name] = a
    return out
