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


from .decode_pool import DecodePool, decode_rgb as _decode_rgb, lease_pool as _lease_pool, release_pool as _release_pool  # noqa: E402,F401


class ImageFolderU8:
    """Iterates `(images uint8 [b,S,S,3] on the device, labels int64 [b])` over [lo, hi) of a FolderIndex."""

    def __init__(self, root_or_index, net, batch_size: int, lo: int = 0, hi: int | None = None,
                 workers: int | None = None):
        self.dataset = root_or_index if isinstance(root_or_index, FolderIndex) else FolderIndex(root_or_index)
        self.net, self.batch_size = net, int(batch_size)
        self.lo, self.hi = lo, (len(self.dataset) if hi is None else hi)
        # decode workers (the reference's DataLoader runs 4 worker processes, utils/train_eval_util.py:49): PROCESSES writing
        # into shared memory (mcm_amd/decode_pool.py) — Pillow decodes under the GIL, threads do not scale.  Default: the
        # node's cores shared between the ranks on it, at most 64 per rank.  workers <= 1: decode in this thread.
        if workers is None:
            from .hostinfo import effective_cpus

            # the CPUs this container may really use (cgroup quota, not os.cpu_count()) shared between the node's ranks; more
            # workers than that run slower (measured on a 16-core quota: 16 workers 12.9k img/s, 64 workers 13.7k, 128 13.4k)
            local_ws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
            workers = int(os.environ.get("MCM_DECODE_WORKERS", 0)) or max(1, min(64, effective_cpus() // local_ws))
        self.workers = int(workers)

    def __len__(self) -> int:
        return max(0, -(-(self.hi - self.lo) // self.batch_size))

    def shard(self, lo: int, hi: int) -> "ImageFolderU8":
        return ImageFolderU8(self.dataset, self.net, self.batch_size, lo, hi, self.workers)

    def close(self):
        """Kept for callers of the earlier per-loader pools: the decode workers are shared by the process now and stop with it."""

    def gather(self, indices):
        """uint8 [b,S,S,3] device batch of the named samples (threshold refinement, mcm_amd/refine.py); more than a handful
        are decoded by the worker processes."""
        import numpy as np
        import torch

        paths = [self.dataset.samples[int(i)][0] for i in indices]
        if os.environ.get("MCM_GPU_JPEG", "0") == "1" and len(paths) >= 8:  # the opt-in device route, as in __iter__
            out = [b.clone() for b in self._jpeg_stream(paths[k:k + self.batch_size] for k in range(0, len(paths), self.batch_size))]
            return out[0] if len(out) == 1 else torch.cat(out)
        if self.workers <= 1 or len(paths) < 32:
            imgs = [np.array(_decode_rgb(p)) for p in paths]  # (np.array: writable — torch warns about read-only arrays)
        else:
            pool, imgs = _lease_pool(self.workers, self.batch_size), []
            try:
                for k in range(0, len(paths), pool.batch):
                    pool.submit(0, paths[k:k + pool.batch])
                    imgs += [a.copy() for a in pool.collect(0)]
            finally:
                _release_pool(pool)
        out = [self.net.resize_crop([torch.from_numpy(a) for a in imgs[k:k + self.net.max_batch]])
               for k in range(0, len(imgs), self.net.max_batch)]
        return out[0] if len(out) == 1 else torch.cat(out)

    def decoded_batches(self, copy: bool = True) -> Iterator:
        """Host side only: `(list of decoded uint8 [H,W,3] arrays, labels int64 [b])` per batch, in dataset order; with
        `workers` > 1 the next two batches are being decoded by the worker processes while the consumer works on this one.
        `copy=False` hands out views of the pool's shared memory, valid until the next batch is requested (what the
        packed pipe wants: it copies them into pinned memory at once)."""
        import torch

        starts = list(range(self.lo, self.hi, self.batch_size))
        chunk_of = lambda s: self.dataset.samples[s:min(s + self.batch_size, self.hi)]  # noqa: E731
        labels_of = lambda s: torch.tensor([t for _, t in chunk_of(s)], dtype=torch.long)  # noqa: E731
        if self.workers <= 1:
            for s in starts:
                yield [_decode_rgb(p) for p, _ in chunk_of(s)], labels_of(s)
            return
        pool = _lease_pool(self.workers, self.batch_size)
        try:
            ahead = pool.slots - 1
            for i in range(min(ahead, len(starts))):
                pool.submit(i % pool.slots, [p for p, _ in chunk_of(starts[i])])
            for i, s in enumerate(starts):
                imgs = pool.collect(i % pool.slots)
                if copy:
                    imgs = [a.copy() for a in imgs]
                yield imgs, labels_of(s)
                # the consumer is done with batch i - that slot's views - once it asks for the next one
                if i + ahead < len(starts):
                    pool.submit((i + ahead) % pool.slots, [p for p, _ in chunk_of(starts[i + ahead])])
        finally:  # also when the consumer stops in the middle of a pass (generator closed): what is in flight is waited out
            _release_pool(pool)

    def _jpeg_stream(self, path_batches) -> Iterator:
        """uint8 [b,S,S,3] device batches for batches of file names through the scorer's JpegFilePipe (kept on the scorer across
        passes and loaders; a second one is made when it is in use)."""
        from .ingest import JpegFilePipe

        pipes = self.net.__dict__.setdefault("_jpeg_pipes", {})
        pipe = pipes.get(self.batch_size)
        if pipe is None or pipe.busy:
            pipe = JpegFilePipe(self.net, self.batch_size, threads=self.workers)
            pipes.setdefault(self.batch_size, pipe)
        pipe.busy = True
        try:
            yield from pipe.stream(path_batches)
        finally:
            pipe.busy = False

    def _iter_jpeg(self) -> Iterator:
        """Files → native entropy decode into pinned memory → coefficients over PCIe → inverse DCT, upsampling, colour, Resize
        + CenterCrop on the device (mcm_amd.ingest.JpegFilePipe); anything that is not a baseline JPEG takes Pillow inside it."""
        import torch

        starts = list(range(self.lo, self.hi, self.batch_size))
        chunk_of = lambda s: self.dataset.samples[s:min(s + self.batch_size, self.hi)]  # noqa: E731
        for i, dev_batch in enumerate(self._jpeg_stream([p for p, _ in chunk_of(s)] for s in starts)):
            yield dev_batch, torch.tensor([t for _, t in chunk_of(starts[i])], dtype=torch.long)

    def __iter__(self) -> Iterator:
        """Default (MCM_GPU_JPEG unset or 0; the CLI's `--decoder pillow`): decode pool (Pillow in worker processes, the
        reference's decoder) → ONE
        packed pinned buffer per batch → ONE asynchronous copy on a copy stream → Resize + CenterCrop on the device
        (mcm_amd.ingest.PackedImagePipe): batch i+1 is decoded, packed and copied while batch i is scored."""
        if os.environ.get("MCM_GPU_JPEG", "0") == "1":
            yield from self._iter_jpeg()
            return
        from .ingest import PackedImagePipe

        labels = []

        def images():
            for imgs, lab in self.decoded_batches(copy=False):
                labels.append(lab)
                yield imgs

        # kept on the scorer across passes and loaders: its pinned slots (grown to the largest batch seen) are expensive to make
        pipes = self.net.__dict__.setdefault("_packed_pipes", {})
        pipe = pipes.get(self.batch_size)
        if pipe is None or getattr(pipe, "busy", False):
            pipe = PackedImagePipe(self.net, self.batch_size, self.batch_size * 512 * 512 * 3 // 2,   # (slots grow on demand)
                                   pack_threads=min(8, max(1, self.workers)))
            pipes.setdefault(self.batch_size, pipe)
        pipe.busy = True
        try:
            for i, dev_batch in enumerate(pipe.stream(images())):
                yield dev_batch, labels[i]
        finally:
            pipe.busy = False
