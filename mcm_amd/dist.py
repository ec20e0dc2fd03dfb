"""Image-sharded data parallelism: one process per GPU, RCCL (backend "nccl") over xGMI.

The path shards as independent units: an image's score depends only on that image and the
replicated prompt bank.  No collective sits on the data path; each dataset
ends with ONE all-gather of the per-rank score shards (≤25 KB/rank — latency-bound), after
which every rank holds the full score vector in the reference's sample order.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np


def world() -> Tuple[int, int]:
    """(rank, world_size) of the initialised default process group, else (0, 1)."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def group_active() -> bool:
    """A default process group exists — of any size.  A 1-rank group (torchrun --nproc-per-node 1) still sends the
    score shards through the collective: that is how the RCCL path is exercised on a 1-GPU box."""
    try:
        import torch.distributed as dist

        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def init_from_env(backend: str | None = None, force: bool = False) -> Tuple[int, int, int]:
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world, local_rank); no-op for 1 process unless `force`
    (a 1-rank process group: every collective of the path then really runs, on RCCL when a GPU is there)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (ws > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if ws == 1:   # a forced 1-rank group: nobody else has to know the port — take a free one (two such processes on one
                import socket   # host used to collide on a fixed default)

                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            else:         # ws > 1 without a launcher-provided port: every rank must agree, so a fixed default it is
                os.environ["MASTER_PORT"] = "29511"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        # gloo's C++ side prints "[Gloo] Rank r is connected ..." on fd 1; callers such as bench.py promise ONE
        # JSON line on stdout, so anything the rendezvous writes goes to stderr instead
        import sys

        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            dist.init_process_group(backend=backend, rank=rank, world_size=ws)
            if backend == "gloo":
                dist.barrier()  # the connection messages are printed lazily, on the first collective
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    return rank, ws, local


def broadcast_tensors(tensors, src: int = 0):
    """Broadcast CPU or device tensors from `src` to every rank, in place (a no-op without a process group).  RCCL moves
    device tensors; CPU tensors take a device bounce under nccl (gloo moves them as they are)."""
    import torch
    import torch.distributed as dist

    if not group_active():
        return tensors
    nccl = dist.get_backend() == "nccl"
    for t in tensors:
        if nccl and not t.is_cuda:
            d = t.cuda()
            dist.broadcast(d, src=src)
            t.copy_(d.cpu())
        elif not nccl and t.is_cuda:
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h.to(t.device))
        else:
            dist.broadcast(t, src=src)
    return tensors


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous index range of `rank`: [r·ceil(n/W), (r+1)·ceil(n/W)) ∩ [0,n).
    Concatenating shards in rank order reproduces the dataset order the reference's
    `shuffle=False` loaders give (utils/train_eval_util.py:96,144)."""
    per = -(-n // world_size)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_gather_scores(local, n_total: int):
    """All-gather per-rank score shards (contiguous `shard_range` split of n_total) into the
    full fp32 vector, on every rank.  `local` is a 1-D torch tensor (device for nccl, CPU
    for gloo) holding this rank's shard."""
    import torch
    import torch.distributed as dist

    rank, ws = world()
    if not group_active():
        return local
    home = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:  # ranks sharing a device (logic checks): host bounce
        local = local.cpu()
    per = -(-n_total // ws)
    buf = torch.zeros(per, dtype=torch.float32, device=local.device)
    buf[: local.numel()] = local.to(torch.float32)
    out = torch.empty(ws * per, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    parts = []
    for r in range(ws):
        lo, hi = shard_range(n_total, r, ws)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts).to(home)


def all_reduce_sum(t):
    """Sum of a float tensor over the ranks, in place, on every rank (a no-op without a process group).  Device tensors go
    over RCCL; under gloo (ranks sharing a device: logic checks) a device tensor takes a host bounce."""
    import torch.distributed as dist

    if not group_active():
        return t
    if dist.get_backend() == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h.to(t.device))
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_gather_histograms(local_scores, edges, net=None):
    """Fixed-bin score histogram summed over ranks (BASELINE.json's "all-gather of per-shard score
    histograms"): a constant-size payload for streaming AUROC estimates.  Exact AUROC/FPR95 parity
    uses `all_gather_scores` (the raw scores are just as cheap).
    Device scores + `net` (a NativeCLIP): the histogram is built by the native kernel
    (`mcm_score_histogram`) and all-reduced over RCCL without leaving HBM.  CPU scores (the gloo test
    path): numpy.histogram + gloo all-reduce."""
    import torch
    import torch.distributed as dist

    rank, ws = world()
    if getattr(local_scores, "is_cuda", False) and net is not None:
        hist = net.histogram(local_scores, edges)
        if group_active():
            if dist.get_backend() == "gloo":  # ranks sharing a device (logic checks): host bounce
                h = hist.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                return h.numpy()
            dist.all_reduce(hist, op=dist.ReduceOp.SUM)
        return hist.cpu().numpy()
    x = local_scores.detach().float().cpu().numpy() if hasattr(local_scores, "detach") else np.asarray(local_scores)
    hist = torch.from_numpy(np.histogram(x, bins=np.asarray(edges, dtype=np.float32))[0].astype(np.int64))
    if group_active():
        backend = dist.get_backend()
        h = hist.cuda() if backend == "nccl" else hist
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        hist = h.cpu()
    return hist.numpy()
