"""Image-folder loader for the real datasets under `--root-dir`: the caller side of SURVEY.md §8f N2.

The reference reads its ImageNet-style sets with `torchvision.datasets.ImageFolder(path, transform=
Resize(224) → CenterCrop(224) → ToTensor → Normalize)` inside a DataLoader with `shuffle=False`
(utils/train_eval_util.py:27-33, 96-146).  Here the host only decodes (Pillow → RGB uint8, any size); the
four transform steps run on the GPU: `NativeCLIP.resize_crop` (mcm_resize_crop_u8, bit-exact against
Pillow's antialiased bilinear resize) produces the uint8 [b,S,S,3] batch and the scoring kernels fold
ToTensor + Normalize into their patch gather (`mcm_score_u8`).  Sample order is ImageFolder's: class
directories sorted by name, files sorted by path inside each class, so score i ↔ sample i as in the
reference.  torchvision itself is not needed (it is not installed in the build image).
"""
from __future__ import annotations

import os
from typing import Iterator, List, Tuple

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


class FolderIndex:
    """`loader.dataset` of the reference's loaders: `__len__`, `.classes`, `.samples`, `.targets`."""

    def __init__(self, root: str):
        self.root = root
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"no class directories under {root}")
        self.samples: List[Tuple[str, int]] = []
        for ci, c in enumerate(self.classes):
            for dirpath, _dirs, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, f), ci))
        if not self.samples:
            raise FileNotFoundError(f"no image files under {root}")
        self.targets = [t for _, t in self.samples]

    def __len__(self) -> int:
        return len(self.samples)

    def first_per_class(self, max_count: int) -> "FolderIndex":
        """The reference's `--subset` rule (utils/train_eval_util.py:56-64): keep, in dataset order, the first
        `max_count` samples of every class."""
        sub = object.__new__(FolderIndex)
        sub.root, sub.classes = self.root, self.classes
        seen, keep = {}, []
        for path, t in self.samples:
            if seen.get(t, 0) < max_count:
                keep.append((path, t))
                seen[t] = seen.get(t, 0) + 1
        sub.samples = keep
        sub.targets = [t for _, t in keep]
        return sub


def _decode_rgb(path):
    import numpy as np
    from PIL import Image

    with Image.open(path) as im:  # torchvision's default loader: PIL, convert("RGB")
        return np.asarray(im.convert("RGB"), dtype=np.uint8).copy()


class ImageFolderU8:
    """Iterates `(images uint8 [b,S,S,3] on the device, labels int64 [b])` over [lo, hi) of a FolderIndex."""

    def __init__(self, root_or_index, net, batch_size: int, lo: int = 0, hi: int | None = None,
                 workers: int | None = None):
        self.dataset = root_or_index if isinstance(root_or_index, FolderIndex) else FolderIndex(root_or_index)
        self.net, self.batch_size = net, int(batch_size)
        self.lo, self.hi = lo, (len(self.dataset) if hi is None else hi)
        # decode pool (the reference's DataLoader runs 4 worker processes, utils/train_eval_util.py:49): Pillow
        # releases the GIL while it decodes, so threads scale; batch i+1 is decoded while batch i is scored
        self.workers = min(32, os.cpu_count() or 4) if workers is None else int(workers)

    def __len__(self) -> int:
        return max(0, -(-(self.hi - self.lo) // self.batch_size))

    def shard(self, lo: int, hi: int) -> "ImageFolderU8":
        return ImageFolderU8(self.dataset, self.net, self.batch_size, lo, hi, self.workers)

    def gather(self, indices):
        """uint8 [b,S,S,3] device batch of the named samples (threshold refinement, mcm_amd/refine.py)."""
        return self.net.resize_crop([__import__("torch").from_numpy(_decode_rgb(self.dataset.samples[int(i)][0])) for i in indices])

    def decoded_batches(self) -> Iterator:
        """Host side only: `(list of decoded uint8 [H,W,3] arrays, labels int64 [b])` per batch, in dataset order; with
        `workers` > 1 batch i+1 is decoded by the pool while the consumer works on batch i."""
        import torch
        from concurrent.futures import ThreadPoolExecutor

        starts = list(range(self.lo, self.hi, self.batch_size))
        chunk_of = lambda s: self.dataset.samples[s:min(s + self.batch_size, self.hi)]  # noqa: E731
        labels_of = lambda s: torch.tensor([t for _, t in chunk_of(s)], dtype=torch.long)  # noqa: E731
        if self.workers <= 1:
            for s in starts:
                yield [_decode_rgb(p) for p, _ in chunk_of(s)], labels_of(s)
            return
        pool = ThreadPoolExecutor(self.workers)
        try:
            submit = lambda s: [pool.submit(_decode_rgb, p) for p, _ in chunk_of(s)]  # noqa: E731
            pending = submit(starts[0]) if starts else None
            for i, s in enumerate(starts):
                futs, pending = pending, (submit(starts[i + 1]) if i + 1 < len(starts) else None)
                yield [f.result() for f in futs], labels_of(s)
        finally:
            pool.shutdown(wait=False, cancel_futures=True)

    def __iter__(self) -> Iterator:
        """Decode pool → ONE packed pinned buffer per batch → ONE asynchronous copy on a copy stream → Resize + CenterCrop on
        the device (mcm_amd.ingest.PackedImagePipe): batch i+1 is decoded, packed and copied while batch i is scored."""
        from .ingest import PackedImagePipe

        labels = []

        def images():
            for imgs, lab in self.decoded_batches():
                labels.append(lab)
                yield imgs

        pipe = PackedImagePipe(self.net, self.batch_size, self.batch_size * 512 * 512 * 3 // 2,   # (slots grow on demand)
                               pack_threads=min(8, max(1, self.workers)))
        for i, dev_batch in enumerate(pipe.stream(images())):
            yield dev_batch, labels[i]
