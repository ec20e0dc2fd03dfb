"""CLIP geometry for the checkpoints the reference accepts, and the ctypes mirror of
`mcm_config` (include/mcm.h).

The checkpoint names are the reference's `--CLIP_ckpt` choices
(eval_ood_detection.py:34-35) and their HF hub ids (utils/train_eval_util.py:19-21);
the dimensions are those of the HF configs (verified by instantiating
`CLIPConfig` in the build container).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, asdict, replace

ABI_VERSION = 4
PREC_BF16 = 0
PREC_F32 = 1
PREC_F16 = 2
DT_F32, DT_F16, DT_BF16 = 0, 1, 2                       # include/mcm.h MCM_DT_*
WEIGHT_OPERANDS = {"auto": 0, "single": 1, "split": 2}  # include/mcm.h MCM_WEIGHTS_*

SCORE_KINDS = {"MCM": 0, "max-logit": 1, "energy": 2, "entropy": 3, "var": 4}


class JpegImage(ctypes.Structure):
    """Field-for-field mirror of `struct mcm_jpeg_image` in include/mcm.h."""

    _fields_ = [("status", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("hs", ctypes.c_int32 * 3), ("vs", ctypes.c_int32 * 3), ("wb", ctypes.c_int32 * 3), ("hb", ctypes.c_int32 * 3),
                ("coef_off", ctypes.c_int64 * 3)]


class CConfig(ctypes.Structure):
    """Field-for-field mirror of `struct mcm_config` in include/mcm.h."""

    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("image_size", ctypes.c_int32),
        ("patch_size", ctypes.c_int32),
        ("v_width", ctypes.c_int32),
        ("v_heads", ctypes.c_int32),
        ("v_layers", ctypes.c_int32),
        ("v_mlp", ctypes.c_int32),
        ("vocab_size", ctypes.c_int32),
        ("max_positions", ctypes.c_int32),
        ("t_width", ctypes.c_int32),
        ("t_heads", ctypes.c_int32),
        ("t_layers", ctypes.c_int32),
        ("t_mlp", ctypes.c_int32),
        ("proj_dim", ctypes.c_int32),
        ("ln_eps", ctypes.c_float),
        ("max_batch", ctypes.c_int32),
        ("max_prompt_tokens", ctypes.c_int32),
        ("weight_operands", ctypes.c_int32),
        ("x2_max_batch", ctypes.c_int32),
    ]


@dataclass(frozen=True)
class ClipGeometry:
    name: str
    image_size: int = 224
    patch_size: int = 16
    v_width: int = 768
    v_heads: int = 12
    v_layers: int = 12
    v_mlp: int = 3072
    vocab_size: int = 49408
    max_positions: int = 77
    t_width: int = 512
    t_heads: int = 8
    t_layers: int = 12
    t_mlp: int = 2048
    proj_dim: int = 512
    ln_eps: float = 1e-5

    @property
    def n_patches(self) -> int:
        g = self.image_size // self.patch_size
        return g * g

    @property
    def v_tokens(self) -> int:
        return self.n_patches + 1

    def vision_flops_per_image(self) -> float:
        """Algorithmic FLOP of the vision tower for one image (SURVEY.md §6:
        L·(8·N·d² + 4·N²·d + 4·N·d·ff) + patch-embed + projection)."""
        n, d, ff, L = self.v_tokens, self.v_width, self.v_mlp, self.v_layers
        enc = L * (8 * n * d * d + 4 * n * n * d + 4 * n * d * ff)
        patch = 2 * self.n_patches * d * 3 * self.patch_size ** 2
        proj = 2 * d * self.proj_dim
        return float(enc + patch + proj)

    def text_flops_per_token(self) -> float:
        d, ff, L = self.t_width, self.t_mlp, self.t_layers
        return float(L * (8 * d * d + 4 * d * ff))

    # ---- batch sizes and the persistent GEMM grid (EXPERIMENTS.md R5.11)
    def gemm_tile_rounds(self, batch: int, cus: int = 256) -> dict:
        """Tile rounds of the persistent 256x256 GEMM kernels at `batch` images: for each full-layer shape
        {qkv, outproj, fc1, fc2} the rounds the launch takes (the fullest XCD decides: M tiles are striped over the 8 XCDs,
        an XCD's cus / 8 workgroups walk its tiles) and the fraction of its tile slots that hold a tile."""
        mt = -(-batch * self.v_tokens // 256)
        per_xcd = cus // 8
        out = {}
        for name, n in (("qkv", 3 * self.v_width), ("outproj", self.v_width), ("fc1", self.v_mlp), ("fc2", self.v_width)):
            nbn = -(-n // 256)
            rounds = max(-(-((mt - x + 7) // 8) * nbn // per_xcd) for x in range(8))
            out[name] = {"rounds": rounds, "fill": mt * nbn / (rounds * cus)}
        return out

    def full_round_batches(self, lo: int = 1, hi: int = 2048, cus: int = 256) -> list:
        """Batch sizes in [lo, hi] at which EVERY vision GEMM fills its last tile round (the largest batch of each such
        M-tile count): per-image throughput peaks there — measured on one MI355X, ViT-B/16 batch 665 / 1330 against 512:
        +2.7 % / +4.1 %; ViT-L/14 255 against 256: +4.4 %; ViT-B/32 1310 against 512: +20 % (profiles/r05_m_*).  Scores do
        not depend on the batch they were computed in (bit-identical kernels), so this is a free choice of the caller."""
        out = []
        mt = 8
        while mt * 256 // self.v_tokens <= hi:
            b = mt * 256 // self.v_tokens  # the largest batch whose token rows fit mt M tiles
            if b >= lo and all(abs(v["fill"] - 1.0) < 1e-9 for v in self.gemm_tile_rounds(b, cus).values()):
                if -(-b * self.v_tokens // 256) == mt:
                    out.append(b)
            mt += 8
        return out

    def to_c(self, *, device: int = 0, precision: int = PREC_BF16, max_batch: int = 512,
             max_prompt_tokens: int = 1024 * 77, weight_operands: int = 0, x2_max_batch: int = 0) -> CConfig:
        d = asdict(self)
        d.pop("name")
        return CConfig(abi_version=ABI_VERSION, device=device, precision=precision,
                       max_batch=max_batch, max_prompt_tokens=max_prompt_tokens,
                       weight_operands=weight_operands, x2_max_batch=x2_max_batch, **d)

    def hf_configs(self):
        """HF `CLIPConfig` of the same geometry (golden-fixture generation and the
        cpu_baseline leg only; transformers is the reference's own dependency)."""
        from transformers import CLIPConfig

        return CLIPConfig(
            text_config=dict(vocab_size=self.vocab_size, hidden_size=self.t_width,
                             intermediate_size=self.t_mlp, num_hidden_layers=self.t_layers,
                             num_attention_heads=self.t_heads,
                             max_position_embeddings=self.max_positions,
                             hidden_act="quick_gelu", layer_norm_eps=self.ln_eps,
                             projection_dim=self.proj_dim, eos_token_id=49407,
                             bos_token_id=49406, pad_token_id=49407),
            vision_config=dict(hidden_size=self.v_width, intermediate_size=self.v_mlp,
                               num_hidden_layers=self.v_layers,
                               num_attention_heads=self.v_heads, image_size=self.image_size,
                               patch_size=self.patch_size, hidden_act="quick_gelu",
                               layer_norm_eps=self.ln_eps, projection_dim=self.proj_dim),
            projection_dim=self.proj_dim,
        )


# --CLIP_ckpt → geometry (reference: eval_ood_detection.py:34-35; hub ids
# utils/train_eval_util.py:19-21)
CHECKPOINTS = {
    "ViT-B/32": ClipGeometry("ViT-B/32", patch_size=32),
    "ViT-B/16": ClipGeometry("ViT-B/16", patch_size=16),
    "ViT-L/14": ClipGeometry("ViT-L/14", patch_size=14, v_width=1024, v_heads=16, v_layers=24,
                             v_mlp=4096, t_width=768, t_heads=12, t_mlp=3072, proj_dim=768),
}
HUB_IDS = {
    "ViT-B/32": "openai/clip-vit-base-patch32",
    "ViT-B/16": "openai/clip-vit-base-patch16",
    "ViT-L/14": "openai/clip-vit-large-patch14",
}

# Reduced geometries for fast parity tests: full-width heads (head_dim 64) and every code
# path of the real towers, few layers / small images so the CPU oracle runs in seconds.
TEST_GEOMETRIES = {
    # 2-layer full-width B/16: per-op / per-layer intermediates
    "B16-2L": replace(CHECKPOINTS["ViT-B/16"], name="B16-2L", v_layers=2, t_layers=2),
    # tiny: 64-px images, 16 patches + CLS = 17 tokens, width 128 (2 heads)
    "tiny": ClipGeometry("tiny", image_size=64, patch_size=16, v_width=128, v_heads=2,
                         v_layers=2, v_mlp=512, vocab_size=49408, max_positions=77,
                         t_width=128, t_heads=2, t_layers=2, t_mlp=512, proj_dim=64),
}


def geometry(name: str) -> ClipGeometry:
    if name in CHECKPOINTS:
        return CHECKPOINTS[name]
    if name in TEST_GEOMETRIES:
        return TEST_GEOMETRIES[name]
    raise KeyError(f"unknown CLIP geometry {name!r}")
