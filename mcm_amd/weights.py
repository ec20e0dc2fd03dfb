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


def synth_state_dict(geo: ClipGeometry, seed: int = 0) -> Dict[str, np.ndarray]:
    return {n: synth_param(n, s, seed) for n, s in param_shapes(geo).items()}


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
