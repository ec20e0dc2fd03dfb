"""NativeCLIP — the `net` object of the reference's hot path, backed by libmcm_hip.so.

Mirrors the duck-typed model contract the reference relies on
(utils/detection_util.py:225,229-230; eval_ood_detection.py:61):

    net.eval()
    net.get_image_features(pixel_values=FloatTensor[b,3,S,S] on device) -> Tensor[b,P]
    net.get_text_features(input_ids=LongTensor[K,S], attention_mask=LongTensor[K,S]) -> Tensor[K,P]

returning real, writable fp32 tensors (the reference calls `.float()` and in-place `/=`
on them), plus the fused path `score_images` that the re-written `get_ood_scores_clip`
uses so the [B,K] softmax never leaves the GPU.

PyTorch is plumbing only: it owns device buffers and the current HIP stream; every FLOP
of the path runs in the hand-written gfx950 kernels behind the C ABI (include/mcm.h).
There is no CPU or eager fallback — a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np

from .config import (ABI_VERSION, CConfig, ClipGeometry, DT_BF16, DT_F16, DT_F32, PREC_BF16, PREC_F16, PREC_F32,
                     SCORE_KINDS, WEIGHT_OPERANDS, geometry)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcm_hip.so")
# the same sources built with -DMCM_HARNESS: A/B kernel arms + mcm_debug_* switches; tests and tools only
HARNESS_LIB_PATH = os.path.join(_HERE, "libmcm_hip_harness.so")
KERNEL_CLASSES = ["patchify", "gemm", "layernorm", "attention", "pool_project", "score", "embed",
                  "gemm_qkv", "gemm_outproj", "gemm_fc1", "gemm_fc2"]  # gemm_*: sub-classes of "gemm" (MCM_KC_GEMM_*)

_libs = {}


class NativeLibraryMissing(RuntimeError):
    pass


def load_library(harness: bool = False):
    """dlopen libmcm_hip.so and declare the C ABI (include/mcm.h).  Raises loudly when the
    extension has not been built — the product path has no fallback.  `harness=True` loads
    libmcm_hip_harness.so instead (A/B tests and tools: extra kernel arms and the mcm_debug_* switches)."""
    if harness in _libs:
        return _libs[harness]
    path = HARNESS_LIB_PATH if harness else LIB_PATH
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C mcm_amd/csrc).  mcm_amd has no CPU fallback.")
    # Both this library and PyTorch link libamdhip64, each from its own ROCm tree.  Whichever is loaded first provides the
    # HIP runtime of the process; when ours (/opt/rocm) came first, torch's later calls failed with "no ROCm-capable device
    # is detected" (measured).  The host code that owns buffers and streams here is torch's, so torch goes first when it
    # is installed; a torch-less process (tests/c_abi) simply runs on /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(path)
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    L.mcm_abi_version.restype = i32
    L.mcm_create.argtypes = [ctypes.POINTER(CConfig), ctypes.POINTER(vp)]
    L.mcm_destroy.argtypes = [vp]
    L.mcm_destroy.restype = None
    L.mcm_last_error.argtypes = [vp]
    L.mcm_last_error.restype = ctypes.c_char_p
    L.mcm_set_weight.argtypes = [vp, ctypes.c_char_p, vp, i32, ctypes.POINTER(ctypes.c_int64), i32]
    L.mcm_weights_operand_exact.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(i32)]
    L.mcm_finalize_weights.argtypes = [vp]
    L.mcm_encode_text.argtypes = [vp, vp, i32, i32, vp, vp]
    L.mcm_encode_image.argtypes = [vp, vp, i32, vp, vp]
    L.mcm_score_features.argtypes = [vp, vp, i32, vp, i32, f32, i32, vp, vp]
    L.mcm_score.argtypes = [vp, vp, i32, vp, i32, f32, i32, vp, vp]
    L.mcm_profile_enable.argtypes = [vp, i32]
    L.mcm_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double),
                                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]
    L.mcm_op_linear.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.mcm_op_linear_ex.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.mcm_op_split_weight.argtypes = [vp, i32, vp, i32, i32, vp, vp]
    L.mcm_op_layernorm.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, f32, i32, vp]
    L.mcm_op_attention.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, vp]
    L.mcm_op_layernorm_split.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, vp]
    L.mcm_op_attention_split.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.mcm_x2_max_batch.argtypes = [vp]
    L.mcm_x2_max_batch.restype = i32
    L.mcm_kernel_faults.argtypes = [vp]
    L.mcm_kernel_faults.restype = i32
    L.mcm_encode_image_x2.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.mcm_score_x2.argtypes = [vp, vp, i32, i32, vp, i32, f32, i32, vp, vp]
    if harness:
        L.mcm_debug_gemm_variant.argtypes = [i32]
        L.mcm_debug_attention_variant.argtypes = [i32]
        L.mcm_debug_attn_spin_budget.argtypes = [ctypes.c_int64]
        L.mcm_debug_clear_faults.argtypes = [vp]
        L.mcm_debug_qkv_chunks.argtypes = [i32]
        L.mcm_debug_ln_fold.argtypes = [i32]
        L.mcm_debug_qkv_head_major.argtypes = [i32]
        L.mcm_debug_ln_tail.argtypes = [i32]
        L.mcm_debug_ln_cluster.argtypes = [i32]
        L.mcm_debug_ln_cluster_spin.argtypes = [i32]
        L.mcm_debug_ln_row.argtypes = [i32]
        L.mcm_debug_ln_cluster_deferred.argtypes = [vp, vp]
        L.mcm_debug_ln_tail_timeouts.argtypes = [vp, vp]
        L.mcm_debug_gemm_dbg.argtypes = [i32]
        L.mcm_debug_nsplit.argtypes = [i32]
        L.mcm_debug_gemm_group_n.argtypes = [i32]
        L.mcm_debug_patch_fold.argtypes = [i32]
        L.mcm_debug_resize_fused_only.argtypes = [i32]
        L.mcm_debug_persistent_grid.argtypes = [i32]
        L.mcm_debug_op_attention.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.mcm_encode_image_u8.argtypes = [vp, vp, i32, vp, vp]
    L.mcm_score_u8.argtypes = [vp, vp, i32, vp, i32, f32, i32, vp, vp]
    L.mcm_reduce_bank.argtypes = [vp, vp, i32, i32, vp, vp]
    L.mcm_resize_crop_u8.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32), i32, vp, vp]
    L.mcm_jpeg_reconstruct.argtypes = [vp, vp, vp, vp, i32, vp, ctypes.POINTER(ctypes.c_int64), vp]
    L.mcm_jpeg_entropy_decode.argtypes = [ctypes.POINTER(ctypes.c_char_p), i32, vp, ctypes.c_int64, vp, vp, i32,
                                          ctypes.POINTER(ctypes.c_int64)]
    L.mcm_pack_u8.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), i32, vp,
                              ctypes.c_int64, i32]
    L.mcm_tokenizer_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(vp)]
    L.mcm_tokenizer_destroy.argtypes = [vp]
    L.mcm_tokenizer_destroy.restype = None
    L.mcm_tokenizer_last_error.argtypes = [vp]
    L.mcm_tokenizer_last_error.restype = ctypes.c_char_p
    L.mcm_tokenizer_vocab_size.argtypes = [vp]
    L.mcm_tokenizer_vocab_size.restype = i32
    L.mcm_tokenizer_encode.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), i32, i32, vp, vp, ctypes.POINTER(i32)]
    L.mcm_encode_image_raw.argtypes = [vp, vp, i32, vp, vp]
    L.mcm_encode_image_ex.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.mcm_encode_text_ex.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.mcm_score_histogram.argtypes = [vp, vp, ctypes.c_int64, vp, i32, vp, vp]
    L.mcm_maha_prepare.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.mcm_maha_score_features.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp]
    L.mcm_measures.argtypes = [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, i32, ctypes.c_double,
                               ctypes.POINTER(ctypes.c_double), vp]
    L.mcm_saturation_check.argtypes = [vp, i32]
    L.mcm_saturation_count.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_uint64), vp]
    if L.mcm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {L.mcm_abi_version()}, this package speaks {ABI_VERSION} — rebuild (make -C mcm_amd/csrc)")
    _libs[harness] = L
    return L


EXPORTED_SYMBOLS = [
    "mcm_abi_version", "mcm_create", "mcm_destroy", "mcm_last_error", "mcm_set_weight",
    "mcm_finalize_weights", "mcm_encode_text", "mcm_encode_image", "mcm_score_features",
    "mcm_score", "mcm_profile_enable", "mcm_profile_read", "mcm_op_linear", "mcm_op_layernorm",
    "mcm_op_attention", "mcm_encode_image_u8", "mcm_score_u8",
    "mcm_reduce_bank", "mcm_measures", "mcm_resize_crop_u8", "mcm_tokenizer_create",
    "mcm_tokenizer_destroy", "mcm_tokenizer_last_error", "mcm_tokenizer_vocab_size", "mcm_tokenizer_encode",
    "mcm_encode_image_raw", "mcm_maha_prepare", "mcm_maha_score_features",
    "mcm_encode_image_ex", "mcm_encode_text_ex", "mcm_score_histogram",
    "mcm_saturation_check", "mcm_saturation_count",
    "mcm_weights_operand_exact", "mcm_op_linear_ex", "mcm_op_split_weight", "mcm_pack_u8",
    "mcm_jpeg_entropy_decode", "mcm_jpeg_reconstruct",
    "mcm_x2_max_batch", "mcm_encode_image_x2", "mcm_score_x2", "mcm_op_layernorm_split", "mcm_op_attention_split",
    "mcm_kernel_faults",
]
HARNESS_ONLY_SYMBOLS = ["mcm_debug_gemm_variant", "mcm_debug_attention_variant", "mcm_debug_qkv_chunks",
                        "mcm_debug_gemm_dbg", "mcm_debug_ln_fold", "mcm_debug_qkv_head_major",
                        "mcm_debug_ln_tail", "mcm_debug_ln_tail_timeouts", "mcm_debug_nsplit",
                        "mcm_debug_gemm_group_n", "mcm_debug_patch_fold", "mcm_debug_resize_fused_only",
                        "mcm_debug_persistent_grid", "mcm_debug_op_attention", "mcm_debug_attn_spin_budget",
                        "mcm_debug_clear_faults", "mcm_debug_ln_cluster", "mcm_debug_ln_cluster_spin",
                        "mcm_debug_ln_cluster_deferred", "mcm_debug_ln_row"]


def _stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class NativeCLIP:
    """CLIP towers + MCM scoring tail on one MI355X."""

    def __init__(self, geo: ClipGeometry | str, state_dict: Dict[str, np.ndarray], *,
                 device: int = 0, precision: str = "fp16", max_batch: int = 512,
                 max_prompt_tokens: int = 1024 * 77, synthetic_weights: Optional[bool] = None,
                 harness: bool = False, weight_operands: str = "auto", x2_max_batch: Optional[int] = None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("NativeCLIP needs a HIP device (no CPU fallback)")
        self.geo = geometry(geo) if isinstance(geo, str) else geo
        self.device = torch.device("cuda", device)
        # "fp16x2": an fp16 handle whose `score_images` / `get_image_features` run the split-activation arm (mcm_score_x2) —
        # every score at fp32-grade accuracy (within one fp32 ulp of the fp32 arm) at about half the fp16 arm's throughput
        self.x2_default = precision == "fp16x2"
        self.precision = {"bf16": PREC_BF16, "fp32": PREC_F32, "f32": PREC_F32, "fp16": PREC_F16,
                          "f16": PREC_F16, "fp16x2": PREC_F16}[precision]
        self.max_batch = int(max_batch)
        # True: seeded stand-in parameters (the hash tokenizer stand-in is then acceptable); False: a real
        # checkpoint (it is refused); None: the caller did not say (mcm_amd.detection falls back to args.weights)
        self.synthetic_weights = synthetic_weights
        self._lib = load_library(harness)  # harness=True: A/B tests and tools only
        torch.cuda.set_device(self.device)
        torch.cuda.init()
        # 16-bit modes: how a GEMM weight is held (include/mcm.h MCM_WEIGHTS_*).  "auto": one 16-bit operand when every
        # weight IS a number of that dtype (the reference's fp16-trained checkpoints), W_hi + W_lo otherwise
        self._cfg = self.geo.to_c(device=device, precision=self.precision, max_batch=max_batch,
                                  max_prompt_tokens=max_prompt_tokens,
                                  weight_operands=WEIGHT_OPERANDS[weight_operands],
                                  # the split-activation workspace (fp16 handles; include/mcm.h mcm_config.x2_max_batch): None /
                                  # 0 = the arm runs at the full batch (twice the activation bytes), n = at most n images per
                                  # x2 call (free up to max_batch / 2), < 0 = none
                                  x2_max_batch=int(x2_max_batch or 0))
        self._h = ctypes.c_void_p()
        rc = self._lib.mcm_create(ctypes.byref(self._cfg), ctypes.byref(self._h))
        if rc:
            raise RuntimeError(f"mcm_create rc={rc}: {self._lib.mcm_last_error(None).decode()}")
        for name, arr in state_dict.items():
            if name == "logit_scale" or name.endswith("position_ids"):
                continue  # unused by MCM (reference utils/detection_util.py:232) / HF buffers
            arr = np.asarray(arr)
            if arr.dtype == np.float16:  # an fp16 checkpoint is handed over as it is (the library widens exactly)
                a, dt = np.ascontiguousarray(arr), DT_F16
            elif str(arr.dtype) == "bfloat16":  # ml_dtypes / safetensors views: raw 16-bit patterns
                a, dt = np.ascontiguousarray(arr).view(np.uint16), DT_BF16
            else:
                a, dt = np.ascontiguousarray(arr, dtype=np.float32), DT_F32
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(self._lib.mcm_set_weight(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), dt,
                                                 shape, a.ndim))
        self._check(self._lib.mcm_finalize_weights(self._h))
        n, sp = ctypes.c_uint64(0), ctypes.c_int32(0)
        self._check(self._lib.mcm_weights_operand_exact(self._h, ctypes.byref(n), ctypes.byref(sp)))
        # vision GEMM-weight elements that are not numbers of the operand dtype, and whether the split-weight
        # GEMMs run (weight_operands="auto": exactly when that count is non-zero)
        self.weights_inexact, self.split_weights = int(n.value), bool(sp.value)

    # -- plumbing ------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc:
            raise RuntimeError(f"libmcm_hip rc={rc}: {self._lib.mcm_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mcm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self):  # reference eval_ood_detection.py:61
        return self

    def _pixels(self, pixel_values):
        import torch

        S = self.geo.image_size
        if pixel_values.dtype == torch.uint8:  # NHWC uint8 ingest (fused /255 + normalise)
            if pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != (S, S, 3):
                raise ValueError(f"uint8 input must be [b,{S},{S},3] NHWC, got {tuple(pixel_values.shape)}")
            return pixel_values.to(device=self.device).contiguous()
        if pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != (3, S, S):
            # HF modeling_clip.py:204-207 raises ValueError on a wrong image size
            raise ValueError(f"Input image size {tuple(pixel_values.shape)} doesn't match model (3x{S}x{S}).")
        return pixel_values.to(device=self.device, dtype=torch.float32).contiguous()

    # -- the model contract ----------------------------------------------------------------
    def get_image_features(self, pixel_values, normalize: bool = False):
        """[b,3,S,S] fp32 (or [b,S,S,3] uint8) → [b,P] fp32.  With the default `normalize=False` this is
        exactly what HF `CLIPModel.get_image_features` returns — the projection output, NOT unit-norm —
        so unmodified reference code that normalises (or, for the Mahalanobis baseline, deliberately
        does not normalise) the rows itself gets what it expects (utils/detection_util.py:158,187,225).
        `normalize=True` fuses the reference's `/= norm` (:226) into the pooling kernel."""
        import torch

        if self.x2_default:
            return self.get_image_features_x2(pixel_values, normalize=normalize)
        px = self._pixels(pixel_values)
        fmt = 1 if px.dtype == torch.uint8 else 0
        out = torch.empty((px.shape[0], self.geo.proj_dim), device=self.device, dtype=torch.float32)
        for s in range(0, px.shape[0], self.max_batch):
            n = min(self.max_batch, px.shape[0] - s)
            self._check(self._lib.mcm_encode_image_ex(self._h, px[s:s + n].data_ptr(), fmt, n, int(bool(normalize)),
                                                      out[s:s + n].data_ptr(), _stream_ptr()))
        return out

    def get_image_features_raw(self, pixel_values):
        """Alias of `get_image_features(pixel_values)` (kept for round-1 callers)."""
        return self.get_image_features(pixel_values, normalize=False)

    def maha_prepare(self, classwise_mean, precision):
        """(means [C,P], precision [P,P]) → opaque state for `maha_scores`."""
        import torch

        mu = classwise_mean.to(device=self.device, dtype=torch.float32).contiguous()
        pr = precision.to(device=self.device, dtype=torch.float32).contiguous()
        C, P = mu.shape
        assert P == self.geo.proj_dim and pr.shape == (P, P)
        w = torch.empty((C, P), device=self.device, dtype=torch.float64)
        k = torch.empty((C,), device=self.device, dtype=torch.float64)
        self._check(self._lib.mcm_maha_prepare(self._h, mu.data_ptr(), pr.data_ptr(), C, w.data_ptr(),
                                               k.data_ptr(), _stream_ptr()))
        return {"prec": pr, "w": w, "k": k, "C": C}

    def maha_scores(self, features, state):
        """features [B,P] fp32 (device) → [B] fp32: min_c 0.5 (f-mu_c) P (f-mu_c)^T."""
        import torch

        f = features.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty(f.shape[0], device=self.device, dtype=torch.float32)
        self._check(self._lib.mcm_maha_score_features(self._h, f.data_ptr(), f.shape[0], state["prec"].data_ptr(),
                                                      state["w"].data_ptr(), state["k"].data_ptr(), state["C"],
                                                      out.data_ptr(), _stream_ptr()))
        return out

    def get_text_features(self, input_ids, attention_mask=None, normalize: bool = False):
        """[K,S] ids → [K,P] fp32: HF `get_text_features` (the text projection output; unit-norm rows
        only with `normalize=True`, which fuses utils/detection_util.py:231).  `attention_mask` is
        accepted for signature parity and ignored: the mask is causal and the pooled row is the first
        EOS, so pads never influence it (golden KAT)."""
        import torch

        ids = np.ascontiguousarray(input_ids.detach().cpu().numpy() if hasattr(input_ids, "detach")
                                   else np.asarray(input_ids), dtype=np.int32)
        if ids.ndim != 2:
            raise ValueError("input_ids must be [K,S]")
        K, S = ids.shape
        if S > self.geo.max_positions:  # HF modeling_clip.py:241-245
            raise ValueError(f"Sequence length must be less than max_position_embeddings (got {S})")
        out = torch.empty((K, self.geo.proj_dim), device=self.device, dtype=torch.float32)
        self._check(self._lib.mcm_encode_text_ex(self._h, ids.ctypes.data_as(ctypes.c_void_p), K, S,
                                                 int(bool(normalize)), out.data_ptr(), _stream_ptr()))
        return out

    # -- fused hot-loop body -----------------------------------------------------------------
    def _bank(self, text_features):
        """The prompt bank as the kernels read it: fp32, contiguous, on this device, [K, proj_dim],
        unit-norm rows are the caller's responsibility (get_text_features(normalize=True))."""
        import torch

        t = text_features.to(device=self.device, dtype=torch.float32).contiguous()
        if t.dim() != 2 or t.shape[1] != self.geo.proj_dim or t.shape[0] == 0:
            raise ValueError(f"text_features must be [K,{self.geo.proj_dim}], got {tuple(t.shape)}")
        return t

    def score_features(self, image_features, text_features, T: float = 1.0, score: str = "MCM"):
        import torch

        f = image_features.to(device=self.device, dtype=torch.float32).contiguous()
        if f.dim() != 2 or f.shape[1] != self.geo.proj_dim:
            raise ValueError(f"image_features must be [b,{self.geo.proj_dim}], got {tuple(f.shape)}")
        t = self._bank(text_features)
        out = torch.empty(f.shape[0], device=self.device, dtype=torch.float32)
        self._check(self._lib.mcm_score_features(self._h, f.data_ptr(), f.shape[0], t.data_ptr(),
                                                 t.shape[0], float(T), SCORE_KINDS[score],
                                                 out.data_ptr(), _stream_ptr()))
        return out

    def score_images(self, pixel_values, text_features, T: float = 1.0, score: str = "MCM", out=None):
        """pixels [b,3,S,S] + pre-encoded bank [K,P] → scores [b] fp32 on device (one
        iteration of the reference loop, utils/detection_util.py:223-248)."""
        import torch

        if self.x2_default:
            return self.score_images_x2(pixel_values, text_features, T, score, out=out)
        px = self._pixels(pixel_values)
        fn = self._lib.mcm_score_u8 if px.dtype == torch.uint8 else self._lib.mcm_score
        t = self._bank(text_features)
        if out is None:
            out = torch.empty(px.shape[0], device=self.device, dtype=torch.float32)
        for s in range(0, px.shape[0], self.max_batch):
            n = min(self.max_batch, px.shape[0] - s)
            self._check(fn(self._h, px[s:s + n].data_ptr(), n, t.data_ptr(), t.shape[0], float(T),
                           SCORE_KINDS[score], out[s:s + n].data_ptr(), _stream_ptr()))
        return out

    # -- split-activation arm: the re-scorer of threshold refinement (include/mcm.h mcm_score_x2) ----------------------
    @property
    def x2_max_batch(self) -> int:
        """Largest batch of the split-activation arm on this handle's workspace (0: not an fp16 handle, or created without it)."""
        return int(self._lib.mcm_x2_max_batch(self._h))

    @property
    def kernel_faults(self) -> int:
        """Non-zero when a persistent kernel gave up a bounded wait (include/mcm.h mcm_kernel_faults): 0 in every correct run."""
        return int(self._lib.mcm_kernel_faults(self._h))

    def score_images_x2(self, pixel_values, text_features, T: float = 1.0, score: str = "MCM", out=None):
        """`score_images` through the split-activation arm: the same weights and workspace, every MFMA operand activation as a
        hi + lo pair of fp16 numbers — scores that agree with the exact-fp32 arm to fp32 round-off, at several times its
        speed.  Any number of images (chunks of `x2_max_batch`)."""
        import torch

        nb = self.x2_max_batch
        if nb <= 0:
            raise RuntimeError("the split-activation arm needs an fp16 handle")
        px = self._pixels(pixel_values)
        fmt = 1 if px.dtype == torch.uint8 else 0  # MCM_PIXELS_U8_NHWC / MCM_PIXELS_F32_NCHW
        t = self._bank(text_features)
        if out is None:
            out = torch.empty(px.shape[0], device=self.device, dtype=torch.float32)
        for s in range(0, px.shape[0], nb):
            n = min(nb, px.shape[0] - s)
            self._check(self._lib.mcm_score_x2(self._h, px[s:s + n].data_ptr(), fmt, n, t.data_ptr(), t.shape[0], float(T),
                                               SCORE_KINDS[score], out[s:s + n].data_ptr(), _stream_ptr()))
        return out

    def get_image_features_x2(self, pixel_values, normalize: bool = False):
        """`get_image_features` through the split-activation arm (raw projected features unless `normalize`)."""
        import torch

        nb = self.x2_max_batch
        if nb <= 0:
            raise RuntimeError("the split-activation arm needs an fp16 handle")
        px = self._pixels(pixel_values)
        fmt = 1 if px.dtype == torch.uint8 else 0
        out = torch.empty((px.shape[0], self.geo.proj_dim), device=self.device, dtype=torch.float32)
        for s in range(0, px.shape[0], nb):
            n = min(nb, px.shape[0] - s)
            self._check(self._lib.mcm_encode_image_x2(self._h, px[s:s + n].data_ptr(), fmt, n, int(bool(normalize)),
                                                      out[s:s + n].data_ptr(), _stream_ptr()))
        return out

    def x2_scorer(self):
        """An object with this scorer's `score_images` signature that runs the split-activation arm (mcm_amd.refine.Rescorer)."""
        return _X2Scorer(self)

    def reduce_bank(self, text_features, K: int, T: int):
        """[K*T,P] unit-norm template features (class-major) → [K,P] ensemble bank."""
        import torch

        f = text_features.to(device=self.device, dtype=torch.float32).contiguous()
        assert f.shape[0] == K * T
        out = torch.empty((K, self.geo.proj_dim), device=self.device, dtype=torch.float32)
        self._check(self._lib.mcm_reduce_bank(self._h, f.data_ptr(), K, T, out.data_ptr(), _stream_ptr()))
        return out

    def resize_crop(self, images):
        """Resize(S) + CenterCrop(S) of the reference's loader transform (utils/train_eval_util.py:27-33)
        on the device: `images` = sequence of uint8 [H_i, W_i, 3] RGB tensors → uint8 [B, S, S, 3], the
        input layout of `score_images` / `get_image_features`.  Bit-exact against Pillow."""
        import torch

        imgs = [im.to(device=self.device, dtype=torch.uint8).contiguous() for im in images]
        B, S = len(imgs), self.geo.image_size
        for im in imgs:
            if im.dim() != 3 or im.shape[2] != 3:
                raise ValueError("images must be uint8 [H, W, 3] RGB")
        ptrs = (ctypes.c_void_p * B)(*[im.data_ptr() for im in imgs])
        hs = (ctypes.c_int32 * B)(*[im.shape[0] for im in imgs])
        ws = (ctypes.c_int32 * B)(*[im.shape[1] for im in imgs])
        out = torch.empty((B, S, S, 3), device=self.device, dtype=torch.uint8)
        self._check(self._lib.mcm_resize_crop_u8(self._h, ptrs, hs, ws, B, out.data_ptr(), _stream_ptr()))
        return out

    def resize_crop_packed(self, packed, offsets, heights, widths, out=None):
        """`resize_crop` over images that already sit back to back in ONE device buffer (`packed`: uint8 1-D device
        tensor; image i = [heights[i], widths[i], 3] at byte offset offsets[i]) — what mcm_amd.ingest.PackedImagePipe
        uploads with one copy per batch.  Asynchronous on the current stream."""
        import torch

        B, S = len(offsets), self.geo.image_size
        if packed.dtype != torch.uint8 or not packed.is_cuda:
            raise ValueError("packed must be a uint8 device tensor")
        base = packed.data_ptr()
        ptrs = (ctypes.c_void_p * B)(*[base + int(o) for o in offsets])
        hs = (ctypes.c_int32 * B)(*[int(v) for v in heights])
        ws = (ctypes.c_int32 * B)(*[int(v) for v in widths])
        if out is None:
            out = torch.empty((B, S, S, 3), device=self.device, dtype=torch.uint8)
        self._check(self._lib.mcm_resize_crop_u8(self._h, ptrs, hs, ws, B, out.data_ptr(), _stream_ptr()))
        return out

    def jpeg_reconstruct(self, coef_dev, meta, quant, n: int, rgb_dev, rgb_offsets):
        """Device half of the JPEG ingest (mcm_jpeg_reconstruct): the coefficients mcm_jpeg_entropy_decode wrote (device
        copy `coef_dev`, uint8 1-D) -> RGB uint8 [H_i, W_i, 3] at rgb_dev + rgb_offsets[i], byte for byte Pillow's decode.
        `meta`: ctypes array of config.JpegImage, `quant`: uint16 numpy [n, 3, 64] (both as the entropy decoder filled them);
        images whose status is not 0 are skipped.  Asynchronous on the current stream."""
        offs = (ctypes.c_int64 * n)(*[int(o) for o in rgb_offsets])
        self._check(self._lib.mcm_jpeg_reconstruct(self._h, coef_dev.data_ptr(), ctypes.addressof(meta), quant.ctypes.data, n,
                                                   rgb_dev.data_ptr(), offs, _stream_ptr()))

    def measures(self, pos_scores, neg_scores, recall_level: float = 0.95, negate: bool = False):
        """`get_measures` (reference utils/detection_util.py:108-119) on device score vectors:
        (auroc, aupr, fpr) with ID = positive class; `negate=True` evaluates on -score, which is
        how `get_and_print_results` (:255) calls it."""
        import torch

        pos = pos_scores.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
        neg = neg_scores.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
        out = (ctypes.c_double * 3)()
        self._check(self._lib.mcm_measures(self._h, pos.data_ptr(), pos.numel(), neg.data_ptr(),
                                           neg.numel(), int(negate), float(recall_level), out,
                                           _stream_ptr()))
        return float(out[0]), float(out[1]), float(out[2])

    def histogram(self, scores, edges):
        """numpy.histogram(scores, edges)[0] on the device → int64 tensor [len(edges)-1]."""
        import torch

        x = scores.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
        e = torch.as_tensor(edges, dtype=torch.float32).to(self.device).contiguous()
        out = torch.empty(e.numel() - 1, device=self.device, dtype=torch.int64)
        self._check(self._lib.mcm_score_histogram(self._h, x.data_ptr(), x.numel(), e.data_ptr(),
                                                  e.numel() - 1, out.data_ptr(), _stream_ptr()))
        return out

    # -- fp16 saturation watch ---------------------------------------------------------------
    def saturation_count(self, reset: bool = True) -> int:
        """Waves that packed an fp16 activation at the saturation value (±65504) since the last reset, over
        every call on this handle; 0 = nothing left the fp16 range.  Always 0 in bf16 / fp32 mode."""
        out = ctypes.c_uint64(0)
        self._check(self._lib.mcm_saturation_count(self._h, int(reset), ctypes.byref(out), _stream_ptr()))
        return int(out.value)

    def saturation_check(self, on: bool):
        self._check(self._lib.mcm_saturation_check(self._h, int(on)))

    def warn_if_saturated(self, what: str = "") -> int:
        """RuntimeWarning when any fp16 activation saturated since the last call (the CLI calls this per set)."""
        n = self.saturation_count(reset=True)
        if n:
            import warnings

            warnings.warn(f"fp16 activations saturated at +-65504 in {n} wave(s){' while scoring ' + what if what else ''}: "
                          "those elements lost precision (outlier channels?) — rerun with --dtype bf16 or fp32 to compare",
                          RuntimeWarning, stacklevel=2)
        return n

    # -- per-kernel timing -------------------------------------------------------------------
    def profile(self, on: bool):
        self._check(self._lib.mcm_profile_enable(self._h, int(on)))

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        n = len(KERNEL_CLASSES)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int64 * n)()
        fl = (ctypes.c_double * n)()
        self._check(self._lib.mcm_profile_read(self._h, ms, cnt, fl))
        return {k: {"ms": ms[i], "launches": int(cnt[i]), "flops": fl[i]}
                for i, k in enumerate(KERNEL_CLASSES)}


class _X2Scorer:
    """`NativeCLIP.x2_scorer()`: the split-activation arm behind the `score_images` / `max_batch` / `device` surface
    mcm_amd.refine.Rescorer drives."""

    label = "split-activation fp16 arm of the same handle (mcm_score_x2)"

    def __init__(self, net):
        self.net, self.device, self.geo = net, net.device, net.geo
        self.max_batch = net.x2_max_batch

    def score_images(self, pixel_values, text_features, T: float = 1.0, score: str = "MCM", out=None):
        return self.net.score_images_x2(pixel_values, text_features, T, score, out=out)


class GraphedScorer:
    """One iteration of the hot loop — `net.score_images(pixels, bank, T, score)` — captured once into a hipGraph and
    replayed: for SMALL batches the ~110 kernel launches of a step cost more host time than the kernels take on the device
    (batch 8 at B/16: 1.4 ms per step eager), a replay is one launch.  The C ABI allocates nothing after `mcm_create`, so the
    call is capturable as it is (include/mcm.h).  The graph holds the pointers it was captured with: `pixels` are copied
    into its static input (device to device) unless the caller fills `scorer.input` itself, the scores land in
    `scorer.output` (valid until the next call).  Per-kernel profiling must be off while capturing.

        scorer = GraphedScorer(net, batch=8, bank=bank)          # capture (one warm-up call + capture)
        scores = scorer(pixels)                                   # replay; same bits as net.score_images(pixels, bank)
    """

    def __init__(self, net: "NativeCLIP", batch: int, bank, T: float = 1.0, score: str = "MCM", uint8: bool = False,
                 input=None):
        import torch

        S = net.geo.image_size
        self.net, self.batch = net, int(batch)
        self.bank = net._bank(bank)
        if input is not None:  # capture on the caller's own static buffer (no copy per call)
            self.input = net._pixels(input)
            if self.input.data_ptr() != input.data_ptr() or self.input.shape[0] != self.batch:
                raise ValueError("input must be a contiguous device tensor of the captured batch size")
        else:
            self.input = (torch.zeros((self.batch, S, S, 3), dtype=torch.uint8, device=net.device) if uint8
                          else torch.zeros((self.batch, 3, S, S), dtype=torch.float32, device=net.device))
        self.output = torch.empty(self.batch, dtype=torch.float32, device=net.device)
        self._stream = torch.cuda.Stream(device=net.device)
        self._stream.wait_stream(torch.cuda.current_stream(net.device))
        with torch.cuda.stream(self._stream):  # first launches set kernel attributes: outside the capture
            net.score_images(self.input, self.bank, T, score, out=self.output)
        self._stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self._stream):
            net.score_images(self.input, self.bank, T, score, out=self.output)

    def __call__(self, pixel_values=None):
        if pixel_values is not None:
            if tuple(pixel_values.shape) != tuple(self.input.shape) or pixel_values.dtype != self.input.dtype:
                raise ValueError(f"the graph was captured for {tuple(self.input.shape)} {self.input.dtype}")
            self.input.copy_(pixel_values, non_blocking=True)
        self.graph.replay()
        return self.output


def build_model(ckpt: str = "ViT-B/16", *, weights: Optional[str] = None, seed: int = 0,
                synthetic_regime: str = "fp16-exact", **kw) -> NativeCLIP:
    """`set_model_clip` counterpart (reference utils/train_eval_util.py:15-36): checkpoint
    name → NativeCLIP.  `weights` = path to a real checkpoint; default = seeded synthetic
    parameters (no checkpoint exists offline), by default rounded to fp16 values like the reference's
    checkpoints are (`synthetic_regime="fp32"`: as drawn — a 16-bit arm then runs the split-weight GEMMs)."""
    from .weights import load_state_dict_file, synth_state_dict

    geo = geometry(ckpt)
    sd = load_state_dict_file(weights, geo) if weights else synth_state_dict(geo, seed, synthetic_regime)
    return NativeCLIP(geo, sd, synthetic_weights=not weights, **kw)
