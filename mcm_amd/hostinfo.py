"""What the host really gives this process: `os.cpu_count()` names the machine's CPUs, but a container is scheduled on its
cgroup's CPU quota (the GPU boxes of this project: 256 CPUs visible, `cpu.max` = 16 cores — 64 decode processes ran slower
than 16, and 128 torch threads are throttled to 16 cores' worth of time)."""
from __future__ import annotations

import math
import os


def cpu_quota():
    """The cgroup CPU quota in cores (float), or None when there is none (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max" and float(p) > 0:
            return float(q) / float(p)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return q / p if q > 0 and p > 0 else None
    except (OSError, ValueError):
        return None


def effective_cpus() -> int:
    """CPUs this process can keep busy: min(affinity mask, cgroup quota), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q is not None:
        n = min(n, max(1, math.floor(q + 1e-9)))
    return max(1, n)
