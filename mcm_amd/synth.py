"""Synthetic inputs honouring the reference's loader contract.

The reference's loaders (utils/train_eval_util.py:87-146) yield `(images, labels)` with
`images` an fp32 `[b,3,S,S]` tensor already normalised, `shuffle=False`; its tokenizer
call (utils/detection_util.py:228) yields `input_ids`/`attention_mask` `[K,S]` padded to
the longest prompt.  No dataset or vocabulary exists in either container, so the build
supplies seeded equivalents:

  pixels  ID : N(0, sigma) noise + a class-conditional low-frequency pattern
          OOD: N(0, sigma) noise + patterns from a disjoint frequency family
          (so ID and OOD score distributions differ and AUROC is non-degenerate)
  tokens  [BOS=49406, r_1..r_n, EOS=49407, pad=49407 ...], n in [3,14], r in [1,49405]
"""
from __future__ import annotations

import math
from typing import Iterator, List, Tuple

import numpy as np

BOS, EOS = 49406, 49407


def make_token_ids(K: int, seed: int = 2, min_len: int = 3, max_len: int = 14,
                   vocab_size: int = 49408) -> Tuple[np.ndarray, np.ndarray]:
    """`tokenizer([...], padding=True)` stand-in → (input_ids, attention_mask) int64 [K,S]."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    lens = rng.integers(min_len, max_len + 1, size=K)
    S = int(lens.max()) + 2
    ids = np.full((K, S), EOS, dtype=np.int64)
    mask = np.zeros((K, S), dtype=np.int64)
    for k in range(K):
        n = int(lens[k])
        ids[k, 0] = BOS
        ids[k, 1:1 + n] = rng.integers(1, min(vocab_size, BOS) - 1, size=n)
        ids[k, 1 + n] = EOS
        mask[k, :n + 2] = 1
    return ids, mask


def _pattern(cls: np.ndarray, size: int, family: int) -> np.ndarray:
    """Class-conditional low-frequency pattern [n,3,size,size] fp32.  `family` selects a
    disjoint set of spatial frequencies (0 = ID, 1 = OOD)."""
    n = cls.shape[0]
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32),
                         indexing="ij")
    out = np.empty((n, 3, size, size), dtype=np.float32)
    for i in range(n):
        c = int(cls[i])
        for ch in range(3):
            fx = (1 + (c * 3 + ch) % 5 + 5 * family) * (2 * math.pi / size)
            fy = (1 + (c * 7 + 2 * ch) % 4 + 4 * family) * (2 * math.pi / size)
            ph = 0.37 * c + 1.1 * ch
            out[i, ch] = np.sin(fx * xx + ph) * np.cos(fy * yy - ph)
    return out


def make_pixels(n: int, size: int, n_classes: int, *, ood: bool, seed: int = 1,
                start: int = 0, noise: float = 1.0, amp: float = 1.5) -> Tuple[np.ndarray, np.ndarray]:
    """Samples [start, start+n) of a deterministic synthetic dataset → (pixels fp32
    [n,3,size,size], labels int64 [n]).  Sample i depends only on (seed, ood, i), so any
    batching / sharding of the index range reproduces the same images."""
    px = np.empty((n, 3, size, size), dtype=np.float32)
    labels = np.empty(n, dtype=np.int64)
    for j in range(n):
        i = start + j
        rng = np.random.Generator(np.random.Philox(key=(seed << 40) | (int(ood) << 39) | i))
        labels[j] = i % n_classes
        px[j] = rng.standard_normal(size=(3, size, size), dtype=np.float32) * np.float32(noise)
    px += np.float32(amp) * _pattern(labels if not ood else labels + 1000, size, int(ood))
    return px, labels


class SyntheticImageSet:
    """`loader.dataset` stand-in: only `__len__` is read by the hot path
    (reference utils/detection_util.py:249)."""

    def __init__(self, n: int, size: int, n_classes: int, ood: bool, seed: int = 1):
        self.n, self.size, self.n_classes, self.ood, self.seed = n, size, n_classes, ood, seed

    def __len__(self) -> int:
        return self.n


class SyntheticLoader:
    """DataLoader stand-in (`shuffle=False`): iterates `(images, labels)` CPU tensors over
    the index range [lo, hi) of a SyntheticImageSet; `lo/hi` give the per-rank shard."""

    def __init__(self, dataset: SyntheticImageSet, batch_size: int, lo: int = 0, hi: int | None = None):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.lo = lo
        self.hi = len(dataset) if hi is None else hi

    def __len__(self) -> int:
        return max(0, -(-(self.hi - self.lo) // self.batch_size))

    def shard(self, lo: int, hi: int) -> "SyntheticLoader":
        """Loader over the sub-range [lo, hi) of the same dataset (per-rank shard)."""
        return SyntheticLoader(self.dataset, self.batch_size, lo, hi)

    def gather(self, indices):
        """Pixels of the named samples (threshold refinement, mcm_amd/refine.py)."""
        import torch

        d = self.dataset
        return torch.from_numpy(np.concatenate([make_pixels(1, d.size, d.n_classes, ood=d.ood, seed=d.seed, start=int(i))[0]
                                                for i in indices]))

    def __iter__(self) -> Iterator:
        import torch

        d = self.dataset
        for s in range(self.lo, self.hi, self.batch_size):
            n = min(self.batch_size, self.hi - s)
            px, lab = make_pixels(n, d.size, d.n_classes, ood=d.ood, seed=d.seed, start=s)
            yield torch.from_numpy(px), torch.from_numpy(lab)


class DeviceNoiseLoader:
    """Bench-scale loader: batches are generated directly in HBM with torch's device
    generator (no host → device copy in the measured path).  Batch `i` is seeded by
    (seed, i) so a run is reproducible on the same device type."""

    def __init__(self, n: int, size: int, batch_size: int, device, seed: int = 1, lo: int = 0,
                 hi: int | None = None):
        self.dataset = SyntheticImageSet(n, size, 1, False, seed)
        self.batch_size, self.device, self.seed = int(batch_size), device, seed
        self.lo, self.hi = lo, (n if hi is None else hi)

    def __len__(self) -> int:
        return max(0, -(-(self.hi - self.lo) // self.batch_size))

    def shard(self, lo: int, hi: int) -> "DeviceNoiseLoader":
        return DeviceNoiseLoader(len(self.dataset), self.dataset.size, self.batch_size, self.device,
                                 self.seed, lo, hi)

    def __iter__(self) -> Iterator:
        import torch

        g = torch.Generator(device=self.device)
        for bi, s in enumerate(range(self.lo, self.hi, self.batch_size)):
            n = min(self.batch_size, self.hi - s)
            g.manual_seed((self.seed << 20) + s)
            x = torch.randn((n, 3, self.dataset.size, self.dataset.size), generator=g,
                            device=self.device, dtype=torch.float32)
            yield x, torch.zeros(n, dtype=torch.long)


class DevicePatternLoader:
    """Headline-scale ID / OOD sets generated in HBM: the same construction as `make_pixels`
    (unit noise + a class-conditional low-frequency pattern from the ID or the OOD frequency
    family) with torch's device generator, so 50k + 10k 224-px images cost no host time and no
    PCIe traffic.  The noise is drawn in aligned blocks of `BLOCK` images, block b seeded by
    (seed, ood, b): image i depends only on (seed, ood, i), so any batching and any sharding of the
    index range — per-rank shards under torchrun included — sees bit-identical pixels."""

    BLOCK = 64

    def __init__(self, n: int, size: int, n_classes: int, batch_size: int, device, *, ood: bool,
                 seed: int = 1, amp: float = 1.5, noise: float = 1.0, tile: float = 0.0,
                 lo: int = 0, hi: int | None = None):
        self.dataset = SyntheticImageSet(n, size, n_classes, ood, seed)
        self.batch_size, self.device = int(batch_size), device
        self.amp, self.noise, self.tile = float(amp), float(noise), float(tile)
        self.lo, self.hi = lo, (n if hi is None else hi)

    def __len__(self) -> int:
        return max(0, -(-(self.hi - self.lo) // self.batch_size))

    def shard(self, lo: int, hi: int) -> "DevicePatternLoader":
        d = self.dataset
        return DevicePatternLoader(d.n, d.size, d.n_classes, self.batch_size, self.device, ood=d.ood,
                                   seed=d.seed, amp=self.amp, noise=self.noise, tile=self.tile, lo=lo, hi=hi)

    def gather(self, indices):
        """Pixels of the named samples (any order) — the same values iteration yields for them (threshold refinement,
        mcm_amd/refine.py).  Generated block by block: every aligned 64-image block that holds a wanted sample once."""
        import torch

        idx = [int(i) for i in indices]
        if not idx:
            return torch.empty((0, 3, self.dataset.size, self.dataset.size), device=self.device)
        out = [None] * len(idx)
        by_block = {}
        for pos, i in enumerate(idx):
            by_block.setdefault(i // self.BLOCK, []).append((pos, i))
        for b, items in sorted(by_block.items()):
            lo = b * self.BLOCK
            hi = min(lo + self.BLOCK, len(self.dataset))
            blk = next(iter(self.shard(lo, hi)._range_batches(self.BLOCK)))[0]
            for pos, i in items:
                out[pos] = blk[i - lo]
        return torch.stack(out)

    def __iter__(self) -> Iterator:
        return self._range_batches(self.batch_size)

    def _range_batches(self, batch_size) -> Iterator:
        import torch

        d, dev = self.dataset, self.device
        S, fam = d.size, int(d.ood)
        g = torch.Generator(device=dev)
        ax = torch.arange(S, device=dev, dtype=torch.float32)
        yy, xx = ax.view(1, 1, S, 1), ax.view(1, 1, 1, S)
        ch = torch.arange(3, device=dev, dtype=torch.float32).view(1, 3, 1, 1)
        w = 2 * math.pi / S
        B = self.BLOCK
        for s in range(self.lo, self.hi, batch_size):
            n = min(batch_size, self.hi - s)
            parts = []
            for b in range(s // B, (s + n - 1) // B + 1):
                g.manual_seed(((d.seed << 24) ^ (fam << 23)) + b)
                blk = torch.randn((B, 3, S, S), generator=g, device=dev, dtype=torch.float32)
                parts.append(blk[max(s - b * B, 0): min(s + n - b * B, B)])
            x = (torch.cat(parts) if len(parts) > 1 else parts[0].clone()) * self.noise
            lab = (torch.arange(s, s + n, device=dev) % d.n_classes)
            c = (lab + (1000 if d.ood else 0)).to(torch.float32).view(n, 1, 1, 1)
            fx = (1 + torch.remainder(c * 3 + ch, 5) + 5 * fam) * w
            fy = (1 + torch.remainder(c * 7 + 2 * ch, 4) + 4 * fam) * w
            ph = 0.37 * c + 1.1 * ch
            x += self.amp * torch.sin(fx * xx + ph) * torch.cos(fy * yy - ph)
            if self.tile:
                # class "texture": a per-channel offset plus a 16-px-periodic pattern, the same in every
                # patch, so (unlike the zero-mean low-frequency pattern) it survives the attention average
                # over patches and gives the pooled CLS feature a class-dependent component — the score
                # spread of a real checkpoint (per-image spread of a few % of |score|) instead of 0.1 %
                dc = torch.sin(1.7 * c + 2.3 * ch + 0.5 * fam)
                tx = torch.sin((2 * math.pi / 16) * (1 + torch.remainder(c, 3)) * xx + 0.9 * c + ch)
                ty = torch.cos((2 * math.pi / 16) * (1 + torch.remainder(c + ch, 2)) * yy - 0.4 * c)
                x += self.tile * (dc + tx * ty)
            yield x, lab


def class_names(K: int) -> List[str]:
    """Concept-bank stand-in for utils/common.py:16-27 (`get_test_labels`): K names."""
    return [f"concept{k:04d}" for k in range(K)]
