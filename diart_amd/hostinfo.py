"""Host-side sizing helpers (no GPU, no torch ops)."""
from __future__ import annotations

import os
from pathlib import Path


def usable_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota.  A GPU
    box can show 256 logical CPUs while the container is granted 16; OpenMP / ATen teams sized
    from the logical count then spin against each other (an 80 000-sample ``torch.stack`` on the
    host was measured at 88 ms instead of 20 us)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def limit_host_threads(max_threads: int = 16) -> int:
    """Cap torch's intra-op host threads at min(usable cores, max_threads); returns the value."""
    import torch
    n = max(1, min(usable_cores(), max_threads, torch.get_num_threads()))
    torch.set_num_threads(n)
    return n
