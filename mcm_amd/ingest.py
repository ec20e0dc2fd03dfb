"""Host → device ingest for the uint8 path (§7 hard part 4: feeding ≥ 10k img/s).

The reference feeds its model through a DataLoader of fp32 [b,3,224,224] tensors and a synchronous `.cuda()` per batch
(utils/detection_util.py:222-223: 602 KB per image over PCIe).  Here the host hands over uint8 pixels — 150 KB per
image for 224² crops — in PINNED memory, one asynchronous copy per batch on a copy stream, double-buffered against the
scoring stream; everything after the copy (Resize + CenterCrop when the images are raw, ToTensor + Normalize fused into
the patch gather) runs on the device.

  PinnedBatchPipe   fixed-size uint8 [B,S,S,3] batches (already cropped on the host, e.g. by a JPEG decoder that resizes)
  PackedImagePipe   variable-size decoded images [H_i,W_i,3]: packed back to back into ONE pinned buffer, ONE copy per
                    batch, then `mcm_resize_crop_u8` over device pointers into the packed buffer (round 3 uploaded every
                    image with its own synchronous `.to(device)`: 512 unpinned copies per batch on the compute stream)
  JpegFilePipe      JPEG FILES: native threads do the bit-serial part of the decode (markers + Huffman,
                    mcm_jpeg_entropy_decode) straight into the pinned upload buffer, the DCT coefficients cross PCIe in place
                    of pixels, and the device does the rest of libjpeg's work (mcm_jpeg_reconstruct: inverse DCT, chroma
                    upsampling, colour conversion — byte for byte Pillow's pixels) before Resize + CenterCrop

PyTorch is plumbing here (pinned allocations, streams, events); no arithmetic.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Sequence, Tuple

import numpy as np


class _Slot:
    def __init__(self, nbytes: int, device):
        import torch

        self.host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.copied = torch.cuda.Event()      # the H2D copy into `dev` has finished
        self.consumed = torch.cuda.Event()    # the compute stream is done reading `dev`
        self.used = False


class _Pipe:
    """`depth` pinned-host / device buffer pairs and a copy stream.  THREE by default: the host fills slot k while the GPU
    scores batch k - 2 and batch k - 1 waits in the queue behind it; with two, filling slot k had to wait for batch k - 2's
    scoring to finish and the GPU idled through every fill (measured: 26.3 instead of 20.4 ms per 512 raw images).  `stage(i)` hands out slot i's pinned host
    view once the compute stream has finished with the slot's previous contents; `push(slot, nbytes)` issues the
    asynchronous copy; `ready(slot)` makes the current (compute) stream wait for it."""

    def __init__(self, net, slot_bytes: int, depth: int = 3):
        import torch

        self.net, self.device = net, net.device
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [_Slot(int(slot_bytes), self.device) for _ in range(max(2, depth))]  # (depth 2 still works, slower)
        self.bytes_copied = 0

    def stage(self, i: int) -> "_Slot":
        s = self.slots[i % len(self.slots)]
        if s.used:
            s.consumed.synchronize()  # host side: do not overwrite pinned memory a copy may still read ...
            s.copied.synchronize()
        return s

    def push(self, s: "_Slot", nbytes: int):
        import torch

        with torch.cuda.stream(self.copy_stream):
            if s.used:
                self.copy_stream.wait_event(s.consumed)  # ... nor device memory the scorer may still read
            s.dev[:nbytes].copy_(s.host[:nbytes], non_blocking=True)
            s.copied.record(self.copy_stream)
        s.used = True
        self.bytes_copied += int(nbytes)

    def ready(self, s: "_Slot"):
        import torch

        torch.cuda.current_stream(self.device).wait_event(s.copied)

    def done(self, s: "_Slot"):
        import torch

        s.consumed.record(torch.cuda.current_stream(self.device))


class PinnedBatchPipe(_Pipe):
    """uint8 [b,S,S,3] host batches → device batches, copy of batch i+1 overlapped with the scoring of batch i.

        pipe = PinnedBatchPipe(net, max_batch)
        for dev_batch in pipe.stream(host_batches):     # host_batches: iterable of uint8 arrays / tensors [b,S,S,3]
            scores = net.score_images(dev_batch, bank)  # on the current stream
    """

    def __init__(self, net, max_batch: int, depth: int = 3):
        S = net.geo.image_size
        self.S, self.max_batch = S, int(max_batch)
        super().__init__(net, self.max_batch * S * S * 3, depth)

    def _push_batch(self, s: _Slot, batch) -> int:
        import torch

        t = batch if hasattr(batch, "is_pinned") else torch.from_numpy(np.ascontiguousarray(batch))
        if t.dtype != torch.uint8 or t.dim() != 4 or tuple(t.shape[1:]) != (self.S, self.S, 3) or t.shape[0] > self.max_batch:
            raise ValueError(f"batches must be uint8 [b<={self.max_batch},{self.S},{self.S},3], got {tuple(t.shape)} {t.dtype}")
        n = t.numel()
        if t.is_pinned() and t.is_contiguous():
            # already in pinned memory (a decoder that writes into pinned buffers): DMA straight out of it; the caller must
            # not overwrite the batch before the copy has run (`copied` of the slot; any later batch of this pipe implies it)
            with torch.cuda.stream(self.copy_stream):
                if s.used:
                    self.copy_stream.wait_event(s.consumed)
                s.dev[:n].copy_(t.reshape(-1), non_blocking=True)
                s.copied.record(self.copy_stream)
            s.used = True
            self.bytes_copied += n
        else:
            s.host[:n].copy_(t.reshape(-1))  # pageable source: one host memcpy into the slot's pinned buffer
            self.push(s, n)
        return t.shape[0]

    def stream(self, host_batches: Iterable) -> Iterator:
        it = iter(host_batches)
        pending = None  # (slot, b) whose copy has been issued
        i = 0
        for batch in it:
            s = self.stage(i)
            b = self._push_batch(s, batch)
            if pending is not None:
                yield from self._emit(pending)
            pending = (s, b)
            i += 1
        if pending is not None:
            yield from self._emit(pending)

    def _emit(self, pending):
        s, b = pending
        self.ready(s)
        yield s.dev[: b * self.S * self.S * 3].view(b, self.S, self.S, 3)
        self.done(s)


class PackedImagePipe(_Pipe):
    """Variable-size decoded RGB images → uint8 [b,S,S,3] device batches (Resize + CenterCrop on the device, bit-exact
    against Pillow: mcm_resize_crop_u8), ONE packed copy per batch.

        pipe = PackedImagePipe(net, max_batch, max_bytes_per_batch)
        for dev_batch in pipe.stream(batches_of_images):   # each item: a list of uint8 [H,W,3] arrays
            scores = net.score_images(dev_batch, bank)
    """

    ALIGN = 16  # every image starts on a 16-byte boundary of the packed buffer

    def __init__(self, net, max_batch: int, max_bytes_per_batch: int, depth: int = 3, pack_threads: int = 1):
        self.max_batch = int(max_batch)
        super().__init__(net, int(max_bytes_per_batch), depth)
        import torch

        # packing a batch is a host memcpy of every image into the pinned buffer (~0.3 GB per 512 raw images): done by
        # `pack_threads` native threads, the work cut by bytes (mcm_pack_u8, csrc/ingest.cpp)
        self.pack_threads = max(1, int(pack_threads))
        self._lib = net._lib

        S = net.geo.image_size
        self.out = [torch.empty((self.max_batch, S, S, 3), dtype=torch.uint8, device=self.device) for _ in self.slots]

    @staticmethod
    def packed_bytes(images: Sequence) -> int:
        a = PackedImagePipe.ALIGN
        return sum((int(np.prod(im.shape)) + a - 1) // a * a for im in images)

    def _fill(self, s: _Slot, images: Sequence) -> Tuple[List[int], List[int], List[int], int]:
        import torch

        if len(images) > self.max_batch:
            raise ValueError(f"{len(images)} images exceed max_batch {self.max_batch}")
        arrs = []
        for im in images:
            a = im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("images must be uint8 [H, W, 3] RGB")
            arrs.append(a)
        need = self.packed_bytes(arrs)
        if need > s.host.numel():  # a batch larger than any seen so far: this slot grows (it is idle: stage() waited)
            grown = _Slot(int(need * 1.25) + (1 << 20), self.device)
            grown.used, grown.consumed, grown.copied = s.used, s.consumed, s.copied
            self.slots[self.slots.index(s)] = grown
            s = grown
        offs, hs, ws, o = [], [], [], 0
        for a in arrs:
            offs.append(o)
            hs.append(a.shape[0])
            ws.append(a.shape[1])
            o += (a.size + self.ALIGN - 1) // self.ALIGN * self.ALIGN

        import ctypes

        arrs = [np.ascontiguousarray(a) for a in arrs]
        n = len(arrs)
        srcs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
        sizes = (ctypes.c_int64 * n)(*[a.size for a in arrs])
        offsets = (ctypes.c_int64 * n)(*offs)
        rc = self._lib.mcm_pack_u8(srcs, sizes, offsets, n, ctypes.c_void_p(s.host.data_ptr()), s.host.numel(), self.pack_threads)
        if rc:
            raise RuntimeError(f"mcm_pack_u8 rc={rc}")
        return s, offs, hs, ws, o

    def stream(self, batches: Iterable[Sequence]) -> Iterator:
        pending = None
        for i, images in enumerate(batches):
            s, offs, hs, ws, nbytes = self._fill(self.stage(i), images)
            self.push(s, nbytes)
            if pending is not None:
                yield from self._emit(*pending)
            pending = (i, s, offs, hs, ws)
        if pending is not None:
            yield from self._emit(*pending)

    def _emit(self, i, s, offs, hs, ws):
        self.ready(s)
        out = self.out[i % len(self.slots)]
        b = len(offs)
        yield self.net.resize_crop_packed(s.dev, offs, hs, ws, out=out[:b])
        self.done(s)


class _JSlot:
    def __init__(self, max_batch: int, coef_bytes: int, device):
        import ctypes

        import torch

        from .config import JpegImage

        self.host = torch.empty(int(coef_bytes), dtype=torch.uint8, pin_memory=True)
        self.dev = torch.empty(0, dtype=torch.uint8, device=device)
        self.rgb = torch.empty(0, dtype=torch.uint8, device=device)
        self.meta = (JpegImage * max_batch)()
        # the same records as int32 columns (status, width, height, ncomp, ...): whole-batch reads without 512 ctypes objects
        self.fields = np.frombuffer(self.meta, dtype=np.int32).reshape(max_batch, ctypes.sizeof(JpegImage) // 4)
        self.quant = np.zeros((max_batch, 3, 64), dtype=np.uint16)
        self.copied, self.consumed = torch.cuda.Event(), torch.cuda.Event()
        self.used = False


class JpegFilePipe:
    """Batches of JPEG file names → uint8 [b,S,S,3] device batches, bit-identical to Pillow decode + Resize + CenterCrop.

        pipe = JpegFilePipe(net, max_batch)
        for dev_batch in pipe.stream(batches_of_paths):   # each item: a list of file names
            scores = net.score_images(dev_batch, bank)

    A producer thread runs the native entropy decoder (`threads` host threads, no GIL) for the next batches into pinned
    slots while this thread uploads, reconstructs, resizes and hands out the current one.  Files the entropy decoder does not
    take (progressive / CMYK JPEGs, PNGs, …) are decoded by Pillow in the producer thread and uploaded as pixels; a file
    nothing can decode raises Pillow's error in the consumer."""

    ALIGN = 16

    def __init__(self, net, max_batch: int, depth: int = 3, threads: int | None = None):
        import torch

        from .hostinfo import effective_cpus

        self.net, self.device, self.max_batch = net, net.device, int(max_batch)
        self.threads = int(threads) if threads else max(1, effective_cpus() - 1)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [_JSlot(self.max_batch, self.max_batch * (640 << 10), self.device) for _ in range(max(2, depth))]
        S = net.geo.image_size
        self.out = [torch.empty((self.max_batch, S, S, 3), dtype=torch.uint8, device=self.device) for _ in self.slots]
        self.bytes_copied = 0
        self.fallback_images = 0
        self.busy = False
        # where the wall clock of a pass goes (seconds, summed over batches): the producer thread waiting for a free slot /
        # decoding; this thread waiting for the producer / preparing and launching a batch
        self.stats = {"batches": 0, "producer_wait_slot_s": 0.0, "producer_decode_s": 0.0, "consumer_wait_s": 0.0, "consumer_launch_s": 0.0}

    def _produce(self, batches, q, stop, free):
        import ctypes
        import os
        import time

        import torch

        from .decode_pool import decode_rgb

        lib = self.net._lib
        try:
            torch.cuda.set_device(self.device)  # (this thread pins memory when a slot grows: on the scorer's device)
            for i, paths in enumerate(batches):
                n = len(paths)
                if n > self.max_batch:
                    raise ValueError(f"{n} files exceed max_batch {self.max_batch}")
                s = self.slots[i % len(self.slots)]
                t0 = time.perf_counter()
                # the slot is this thread's only once the consumer has handed it back (after its `consumed` event is recorded:
                # ADVICE r4 — with the queue alone a 2-slot pipe re-entered slot 0 while batch 0 was still being uploaded)
                while not free[i % len(self.slots)].acquire(timeout=0.2):
                    if stop.is_set():
                        return
                if s.used:
                    s.consumed.synchronize()
                    s.copied.synchronize()
                t1 = time.perf_counter()
                arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
                used = ctypes.c_int64(0)
                while True:
                    rc = lib.mcm_jpeg_entropy_decode(arr, n, s.host.data_ptr(), s.host.numel(), s.meta, s.quant.ctypes.data,
                                                     self.threads, ctypes.byref(used))
                    if rc == 0:
                        break
                    if rc != -7 or used.value <= s.host.numel():  # MCM_ERANGE = the slot is too small: grow and retry
                        raise RuntimeError(f"mcm_jpeg_entropy_decode rc={rc}")
                    s.host = torch.empty(int(used.value * 1.25) + (1 << 20), dtype=torch.uint8, pin_memory=True)
                fb = np.nonzero(s.fields[:n, 0])[0].tolist()  # status != 0
                if len(fb) >= 4 and self.threads > 1:  # several files for Pillow (progressive JPEGs, PNGs): its worker processes
                    from .decode_pool import lease_pool, release_pool

                    pool = lease_pool(min(self.threads, 16), self.max_batch)
                    try:
                        pool.submit(0, [paths[j] for j in fb])
                        fallbacks = {j: a.copy() for j, a in zip(fb, pool.collect(0))}
                    finally:
                        release_pool(pool)
                else:
                    fallbacks = {j: decode_rgb(paths[j]) for j in fb}
                t2 = time.perf_counter()
                st = self.stats
                st["batches"] += 1
                st["producer_wait_slot_s"] += t1 - t0
                st["producer_decode_s"] += t2 - t1
                item = (i, s, n, int(used.value), fallbacks)
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.2)
                        break
                    except Exception:
                        continue
                if stop.is_set():
                    return
            q.put(None)
        except BaseException as e:  # handed to the consumer
            q.put(e)

    def stream(self, batches: Iterable[Sequence]) -> Iterator:
        import queue
        import threading
        import time

        import torch

        if getattr(self, "_streaming", False):
            raise RuntimeError("JpegFilePipe.stream: the previous pass over this pipe is still open (close its generator first)")
        self._streaming = True
        q = queue.Queue(maxsize=len(self.slots))
        stop = threading.Event()
        free = [threading.Semaphore(1) for _ in self.slots]  # slot k may be (re)filled by the producer
        th = threading.Thread(target=self._produce, args=(iter(batches), q, stop, free), daemon=True)
        th.start()
        try:
            while True:
                tq = time.perf_counter()
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                tg = time.perf_counter()
                self.stats["consumer_wait_s"] += tg - tq
                i, s, n, nbytes, fallbacks = item
                wsa, hsa = s.fields[:n, 1].astype(np.int64), s.fields[:n, 2].astype(np.int64)  # (width, height) as decoded
                for j, a in fallbacks.items():
                    hsa[j], wsa[j] = a.shape[:2]
                sizes = (hsa * wsa * 3 + self.ALIGN - 1) // self.ALIGN * self.ALIGN
                offs = (np.cumsum(sizes) - sizes).tolist()
                hs, ws, o = hsa.tolist(), wsa.tolist(), int(sizes.sum())
                if s.dev.numel() < nbytes or s.rgb.numel() < o:  # this slot's device buffers grow to the largest batch seen
                    if s.used:
                        s.consumed.synchronize()
                    if s.dev.numel() < nbytes:
                        s.dev = torch.empty(int(nbytes * 1.25) + (1 << 20), dtype=torch.uint8, device=self.device)
                    if s.rgb.numel() < o:
                        s.rgb = torch.empty(int(o * 1.25) + (1 << 20), dtype=torch.uint8, device=self.device)
                with torch.cuda.stream(self.copy_stream):
                    if s.used:
                        self.copy_stream.wait_event(s.consumed)
                    if nbytes:
                        s.dev[:nbytes].copy_(s.host[:nbytes], non_blocking=True)
                    for j, a in fallbacks.items():  # (rare; pageable pixels, a copy each)
                        s.rgb[offs[j]: offs[j] + a.size].copy_(torch.from_numpy(np.array(a)).reshape(-1))
                    s.copied.record(self.copy_stream)
                s.used = True
                self.bytes_copied += nbytes
                self.fallback_images += len(fallbacks)
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(s.copied)
                if len(fallbacks) < n:
                    self.net.jpeg_reconstruct(s.dev, s.meta, s.quant, n, s.rgb, offs)
                out = self.net.resize_crop_packed(s.rgb, offs, hs, ws, out=self.out[i % len(self.slots)][:n])
                self.stats["consumer_launch_s"] += time.perf_counter() - tg
                try:
                    yield out
                finally:  # also when the generator is closed at the yield: the slot's last reader is recorded either way
                    s.consumed.record(torch.cuda.current_stream(self.device))
                    free[i % len(self.slots)].release()
        finally:
            stop.set()
            try:  # a producer blocked in q.put() sees `stop` within its timeout; one blocked on a slot within 0.2 s
                while True:
                    q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=30.0)
            self._streaming = False
