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


def load_tokenizer(ckpt: str):
    """Real CLIP BPE tokenizer when its vocabulary is available locally, else HashTokenizer.
    transformers 5.x returns an EMPTY-vocabulary tokenizer instead of raising when the files
    are missing (every prompt then tokenises to the same ids), so the result is validated."""
    try:
        from transformers import CLIPTokenizer

        tok = CLIPTokenizer.from_pretrained(ckpt, local_files_only=True)
        probe = tok(["a photo of a tench", "a photo of a goldfish"], padding=True, return_tensors="np")
        ids = np.asarray(probe["input_ids"])
        if len(tok) >= 49408 and ids[0, 0] == BOS and ids.max() == EOS and (ids[0] != ids[1]).any():
            return tok
    except Exception:
        pass
    return HashTokenizer()
