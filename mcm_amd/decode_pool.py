"""Decode worker PROCESSES writing into shared memory: the host half of the image-folder loader.

Pillow holds the GIL while it decodes a JPEG (measured: 410 – 480 img/s with 1, 2, 4 or 8 threads; 2 000 img/s on the GPU
box with 32), so a thread pool feeds a few hundred images a second into a scorer that takes 24 000.  The reference's
DataLoader uses worker processes for the same reason (utils/train_eval_util.py:49) and pickles every decoded batch through
a pipe.  Here:

* the workers are plain `python -m mcm_amd.decode_pool` children started with subprocess (vfork + exec): they import numpy
  and Pillow and nothing else.  NOT multiprocessing: its fork start method copies the page tables of the parent — a process
  with a HIP context maps hundreds of GB, one fork took 0.7 s, 64 workers 48 s (measured) — and its spawn / forkserver
  methods re-import the parent's main module (torch) in every child;
* image j of a batch is written at j * stride of a slot of ONE shared mapping (a file in /dev/shm, or in the temp directory
  when /dev/shm is missing or too small; unlinked as soon as the workers have opened it) and the parent hands out numpy
  views of it: nothing but file names (parent -> worker, one pipe per worker, tasks of 8 images dealt round-robin) and
  8-byte completion records (workers -> parent, one shared pipe; writes below PIPE_BUF are atomic) crosses a pipe;
* `slots` batches can be in flight; the views of a batch stay valid until `slots - 1` further batches have been submitted.
"""
from __future__ import annotations

import json
import mmap
import os
import struct
import sys


def decode_rgb(path):
    """torchvision's default loader (PIL, convert("RGB")) -> uint8 [H, W, 3]; a read-only array with its own buffer."""
    import numpy as np
    from PIL import Image

    with Image.open(path) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")  # (an RGB file is not copied a second time)
        return np.asarray(im, dtype=np.uint8)  # Pillow's array interface hands out a copy of the pixels


def _views(buf, slots, batch, stride):
    import numpy as np

    pix = np.frombuffer(buf, dtype=np.uint8, count=slots * batch * stride)
    meta = np.frombuffer(buf, dtype=np.int32, count=slots * batch * 3, offset=slots * batch * stride).reshape(-1, 3)
    return pix, meta


def _worker_main(argv):
    path, slots, batch, stride, done_fd = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4])
    fd = os.open(path, os.O_RDWR)
    buf = mmap.mmap(fd, 0)
    os.close(fd)
    pix, meta = _views(buf, slots, batch, stride)
    os.write(done_fd, struct.pack("ii", -1, 0))  # "mapped": the parent may unlink the file
    for line in sys.stdin:
        t = json.loads(line)
        slot, j0, paths = t["slot"], t["j0"], t["paths"]
        for k, p in enumerate(paths):
            j = slot * batch + j0 + k
            try:
                a = decode_rgb(p)
                if a.nbytes <= stride:
                    pix[j * stride: j * stride + a.nbytes] = a.reshape(-1)
                    meta[j] = (a.shape[0], a.shape[1], 1)
                else:
                    meta[j] = (a.shape[0], a.shape[1], 2)   # larger than a place: the parent decodes it itself
            except Exception:
                meta[j] = (0, 0, 3)                          # the parent decodes it itself and raises the real error
        os.write(done_fd, struct.pack("ii", slot, len(paths)))


class DecodePool:
    TASK = 8  # images per task

    def __init__(self, workers: int, batch: int, slots: int = 3, stride: int = 3 << 20):
        import subprocess
        import tempfile
        import weakref

        self.batch, self.slots, self.stride = int(batch), int(slots), int(stride)
        size = self.slots * self.batch * (self.stride + 12)
        size = (size + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
        where = tempfile.gettempdir()
        try:  # tmpfs pages exist once written; a mount smaller than the worst case would SIGBUS a worker
            st = os.statvfs("/dev/shm")
            if st.f_bavail * st.f_frsize >= size:
                where = "/dev/shm"
        except OSError:
            pass
        fd, path = tempfile.mkstemp(prefix="mcm_decode_", dir=where)
        try:
            os.ftruncate(fd, size)
            self.buf = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        self.pix, self.meta = _views(self.buf, self.slots, self.batch, self.stride)
        self.done_r, done_w = os.pipe()
        env = dict(os.environ)
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        cmd = [sys.executable, "-m", "mcm_amd.decode_pool", path, str(self.slots), str(self.batch), str(self.stride), str(done_w)]
        self.procs = []
        try:
            for _ in range(max(1, int(workers))):
                self.procs.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, pass_fds=(done_w,), env=env, text=True, bufsize=1))
        finally:
            os.close(done_w)
        self._fin = weakref.finalize(self, DecodePool._shutdown, self.procs, self.done_r, path)
        self.busy = False  # leased to a loader (mcm_amd/folder.py)
        self._pending = [0] * self.slots
        self._paths = [None] * self.slots
        self._next = 0
        self._mapped = 0
        self._path = path
        while self._mapped < len(self.procs):  # every worker has the file open: drop its name
            self._read_done()
        try:
            os.unlink(path)
        except OSError:
            pass

    @staticmethod
    def _shutdown(procs, done_r, path):
        for p in procs:
            try:
                p.stdin.close()  # EOF ends the worker's loop
            except Exception:
                pass
        for p in procs:
            try:
                p.wait(timeout=2)
            except Exception:
                p.kill()
        try:
            os.close(done_r)
        except OSError:
            pass
        try:
            os.unlink(path)
        except OSError:
            pass

    def close(self):
        self._fin()

    def alive(self) -> bool:
        return self._fin.alive and all(p.poll() is None for p in self.procs)

    def _read_done(self):
        import select

        while not select.select([self.done_r], [], [], 1.0)[0]:
            dead = [p.returncode for p in self.procs if p.poll() is not None]
            if dead:
                raise RuntimeError(f"a decode worker process exited (return codes {dead})")
        rec = os.read(self.done_r, 8)
        while len(rec) < 8:
            rec += os.read(self.done_r, 8 - len(rec))
        s, n = struct.unpack("ii", rec)
        if s < 0:
            self._mapped += 1
        else:
            self._pending[s] -= n

    def drain(self):
        """Waits out whatever is still in flight (a consumer that stopped in the middle of a pass)."""
        for slot in range(self.slots):
            while self._pending[slot] > 0:
                self._read_done()

    def submit(self, slot: int, paths):
        assert self._pending[slot] == 0 and len(paths) <= self.batch
        self._paths[slot] = list(paths)
        self._pending[slot] = len(paths)
        for j0 in range(0, len(paths), self.TASK):
            p = self.procs[self._next % len(self.procs)]
            self._next += 1
            p.stdin.write(json.dumps({"slot": slot, "j0": j0, "paths": self._paths[slot][j0:j0 + self.TASK]}) + "\n")
            p.stdin.flush()

    def collect(self, slot: int):
        """Blocks until every image of the batch in `slot` is decoded; list of uint8 [H,W,3] views of the shared slot."""
        while self._pending[slot] > 0:
            self._read_done()
        out = []
        for k, path in enumerate(self._paths[slot]):
            j = slot * self.batch + k
            h, w, st = (int(v) for v in self.meta[j])
            if st == 1:
                out.append(self.pix[j * self.stride: j * self.stride + h * w * 3].reshape(h, w, 3))
            else:
                out.append(decode_rgb(path))  # too large for a place, or failed in the worker: here the error surfaces
        return out


# One pool per (workers, batch) serves every loader of the process: the CLI walks five datasets one after the other, and
# starting the workers per dataset cost more than scoring the smaller sets.  A caller that finds the shared pool in use makes
# its own (closed on release).
_POOLS: dict = {}


def lease_pool(workers: int, batch: int) -> DecodePool:
    p = _POOLS.get((workers, batch))
    if p is None or not p.alive():
        p = _POOLS[(workers, batch)] = DecodePool(workers, batch)
    if p.busy:
        p = DecodePool(workers, batch)
        p.private = True
    p.busy = True
    p.drain()
    return p


def release_pool(p: DecodePool):
    p.drain()
    p.busy = False
    if getattr(p, "private", False):
        p.close()


if __name__ == "__main__":
    _worker_main(sys.argv[1:])
