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


def parse_cpulist(text: str) -> list:
    """"0-3,8,10-11" (sysfs cpulist format) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(device_index: int) -> list:
    """CPUs on the NUMA node the GPU hangs off (sysfs ``local_cpulist`` of its PCI function), [] if unknown."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        return parse_cpulist(Path(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read_text())
    except Exception:
        return []


def rank_cpu_share(local_rank: int, local_world: int, allowed, local_of=gpu_local_cpus,
                   device_of=lambda r: r) -> list:
    """The cores rank ``local_rank`` of ``local_world`` should pin itself (launching thread + worker pool) to:
    the allowed cores nearest its GPU, split evenly between the ranks whose GPUs share those cores — every rank
    computes the same partition from sysfs alone, no communication.  [] = leave the affinity as it is (unknown
    topology, or fewer near cores than ranks)."""
    allowed = sorted(allowed)
    near = {}
    for r in range(local_world):
        cpus = [c for c in local_of(device_of(r)) if c in set(allowed)]
        near[r] = tuple(cpus) if cpus else tuple(allowed)
    mine = near[local_rank]
    peers = [r for r in range(local_world) if near[r] == mine]
    share = len(mine) // len(peers)
    if share < 1:
        return []
    i = peers.index(local_rank)
    return list(mine[i * share:(i + 1) * share])


def bind_rank(local_rank: int, local_world: int, device_index=None) -> dict:
    """Pin this process to its share of the cores nearest its GPU (``rank_cpu_share``).  The reference's
    ``Parallelize`` leaves placement to the OS (/root/reference/src/diart/inference.py:526-559); with 8 ranks
    that time-share a 16-core grant, a rank whose pool migrates across sockets pays remote-memory latency on
    every pinned-buffer touch.  Never raises; returns what it did (goes into the bench line)."""
    info = {"bound": False, "cpus": None}
    from .config import setting
    if local_world <= 1 or str(setting("affinity", None, "1")) == "0" or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        allowed = os.sched_getaffinity(0)
        fixed = device_index if device_index is not None else None
        cpus = rank_cpu_share(local_rank, local_world, allowed,
                              device_of=(lambda r: fixed) if fixed is not None else (lambda r: r))
        if cpus:
            os.sched_setaffinity(0, cpus)
            info = {"bound": True, "cpus": f"{cpus[0]}-{cpus[-1]}" if cpus == list(range(cpus[0], cpus[-1] + 1)) else cpus,
                    "n": len(cpus)}
    except Exception as exc:      # noqa: BLE001 — placement is an optimisation, never a failure
        info["error"] = repr(exc)[:80]
    return info
