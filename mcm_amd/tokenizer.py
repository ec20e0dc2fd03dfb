"""Prompt → token ids.  Host-side string work, outside the kernel scope (SURVEY.md §2 #5,
§8f N4): token ids are an *input* to the native path.

The reference calls `CLIPTokenizer.from_pretrained(args.ckpt)` and
`tokenizer(list[str], padding=True, return_tensors="pt")` (utils/detection_util.py:216,228).
The BPE vocabulary files are absent from both containers (hub offline), so when the real
tokenizer cannot be loaded a deterministic stand-in honouring the same call contract is
used: BOS 49406, one id per whitespace-separated word (stable hash into [1, 49405]),
EOS 49407, right-padded with 49407 to the longest prompt.
"""
from __future__ import annotations

import zlib
from typing import Dict, List

import numpy as np

BOS, EOS = 49406, 49407


class HashTokenizer:
    max_len = 77

    def __call__(self, texts: List[str], padding: bool = True, return_tensors: str = "pt") -> Dict:
        rows = []
        for t in texts:
            words = t.lower().split()[: self.max_len - 2]
            rows.append([BOS] + [1 + zlib.crc32(w.encode()) % (BOS - 1) for w in words] + [EOS])
        S = max(len(r) for r in rows)
        ids = np.full((len(rows), S), EOS, dtype=np.int64)
        mask = np.zeros((len(rows), S), dtype=np.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
            mask[i, :len(r)] = 1
        if return_tensors == "pt":
            import torch

            return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        return {"input_ids": ids, "attention_mask": mask}


class NativeBPETokenizer:
    """The library's C++ CLIP BPE (mcm_amd/csrc/tokenizer.cpp) behind the call contract the
    reference uses: `tok(list[str], padding=True, return_tensors="pt")` → input_ids, attention_mask."""
    max_len = 77

    def __init__(self, vocab_json: str, merges_txt: str):
        import ctypes

        from .engine import load_library

        self._lib = load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.mcm_tokenizer_create(vocab_json.encode(), merges_txt.encode(), ctypes.byref(self._h))
        if rc:
            raise RuntimeError(f"mcm_tokenizer_create rc={rc}: {self._lib.mcm_tokenizer_last_error(None).decode()}")

    def __len__(self):
        return int(self._lib.mcm_tokenizer_vocab_size(self._h))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mcm_tokenizer_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def __call__(self, texts: List[str], padding: bool = True, return_tensors: str = "pt",
                 capacity: int = 4096) -> Dict:
        import ctypes

        if isinstance(texts, str):
            texts = [texts]
        n = len(texts)
        arr = (ctypes.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        ids = np.empty((n, capacity), dtype=np.int32)
        mask = np.empty((n, capacity), dtype=np.int32)
        S = ctypes.c_int32()
        rc = self._lib.mcm_tokenizer_encode(self._h, arr, n, capacity, ids.ctypes.data, mask.ctypes.data,
                                            ctypes.byref(S))
        if rc:
            raise RuntimeError(f"mcm_tokenizer_encode rc={rc}: {self._lib.mcm_tokenizer_last_error(self._h).decode()}")
        S = S.value  # the C side writes rows of length S contiguously
        ids = ids.reshape(-1)[: n * S].reshape(n, S).astype(np.int64)
        mask = mask.reshape(-1)[: n * S].reshape(n, S).astype(np.int64)
        if not padding:
            return {"input_ids": [r[m.astype(bool)].tolist() for r, m in zip(ids, mask)],
                    "attention_mask": [m[m.astype(bool)].tolist() for m in mask]}
        if return_tensors == "pt":
            import torch

            return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        return {"input_ids": ids, "attention_mask": mask}


class TokenizerUnavailable(RuntimeError):
    pass


def load_tokenizer(ckpt: str, *, allow_hash: bool = True):
    """`CLIPTokenizer.from_pretrained(args.ckpt)` of the reference (utils/detection_util.py:216), in order:
      1. the native C++ BPE when `ckpt` is a directory holding vocab.json + merges.txt;
      2. HF's CLIPTokenizer when its vocabulary is available locally (`ckpt` = hub id or directory).
         transformers 5.x returns an EMPTY-vocabulary tokenizer instead of raising when the files are
         missing (every prompt then tokenises to the same ids), so the result is validated;
      3. HashTokenizer — only with `allow_hash`, and never silently: a RuntimeWarning says that the
         ids are stand-ins (fine for synthetic weights, meaningless with a real checkpoint).
    `allow_hash=False` (the CLI sets it when --weights is given) raises TokenizerUnavailable instead."""
    import os
    import warnings

    why = []
    if ckpt and os.path.isdir(ckpt):
        v, m = os.path.join(ckpt, "vocab.json"), os.path.join(ckpt, "merges.txt")
        if os.path.exists(v) and os.path.exists(m):
            return NativeBPETokenizer(v, m)
        why.append(f"{ckpt} holds no vocab.json + merges.txt")
    try:
        from transformers import CLIPTokenizer

        tok = CLIPTokenizer.from_pretrained(ckpt, local_files_only=True)
        probe = tok(["a photo of a tench", "a photo of a goldfish"], padding=True, return_tensors="np")
        ids = np.asarray(probe["input_ids"])
        if len(tok) >= 49408 and ids[0, 0] == BOS and ids.max() == EOS and (ids[0] != ids[1]).any():
            return tok
        why.append(f"transformers returned an empty-vocabulary tokenizer for {ckpt!r}")
    except Exception as e:  # offline hub, unknown id, transformers missing
        why.append(f"CLIPTokenizer.from_pretrained({ckpt!r}) failed: {type(e).__name__}: {e}".splitlines()[0])
    msg = "no CLIP BPE vocabulary available (" + "; ".join(why) + ")"
    if not allow_hash:
        raise TokenizerUnavailable(msg + "; pass --tokenizer-dir <dir with vocab.json, merges.txt>")
    warnings.warn(msg + ": using HashTokenizer stand-in ids — valid only with synthetic weights",
                  RuntimeWarning, stacklevel=2)
    return HashTokenizer()
