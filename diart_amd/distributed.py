"""One process per GPU; streams are independent, so the only collective on this path is the
one-time broadcast of the packed weights from rank 0 (RCCL over xGMI on a GPU node, gloo in the
CPU tests).  The reference's ``Parallelize`` gives every worker its own full model copy loaded
from disk/hub (``/root/reference/src/diart/inference.py:484-493``, "TODO share models across
processes"); here rank 0 loads (or synthesises) the state dict once and ships 23 MB.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the default
    process group when WORLD_SIZE > 1 (backend: nccl (= RCCL) with a GPU, else gloo)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:   # DZ_DIST_BACKEND=gloo: rehearse the multi-rank path on ONE GPU
            backend = os.environ.get("DZ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("DZ_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def torchrun_command(n: int, script: str, argv: Sequence[str], port: Optional[int] = None) -> List[str]:
    """The command the driver itself uses for N > 1 (one rank per GPU, single node, 127.0.0.1)."""
    import sys
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def self_launch(n: int, script: str, argv: Sequence[str]) -> Optional[int]:
    """``script --gpus N`` started as ONE process (no WORLD_SIZE in the environment) starts its own
    N ranks, the way the reference's ``Parallelize`` spawns its pool from inside the call
    (``/root/reference/src/diart/inference.py:526-559``): re-runs the same command line under
    ``torch.distributed.run`` — one rank per GPU, RCCL — and returns the launcher's exit code.
    Returns None when there is nothing to do (N == 1, or this process already is a rank).

    Fails loudly when the node shows fewer than N GPUs: N ranks squeezed onto fewer devices would
    print an N-GPU line for a job that never had N GPUs.  The single-GPU rehearsal of the multi-rank
    code path asks for exactly that, explicitly: ``DZ_FORCE_DEVICE=<ordinal>`` (every rank on that
    GPU; RCCL refuses two ranks on one device, so the rehearsal needs ``DZ_DIST_BACKEND=gloo``)."""
    import subprocess
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return None
    have = torch.cuda.device_count()
    if have < n and "DZ_FORCE_DEVICE" not in os.environ:
        raise SystemExit(f"{script}: --gpus {n} but this node shows {have} GPU(s); refusing to run {n} ranks on "
                         f"fewer devices (single-GPU rehearsal: DZ_FORCE_DEVICE=0 DZ_DIST_BACKEND=gloo)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")      # torchrun would set (and warn about) it anyway
    return subprocess.call(torchrun_command(n, script, argv), env=env)


def broadcast_state(state: Optional[Dict[str, torch.Tensor]], spec: Sequence[Tuple[str, Tuple[int, ...], torch.dtype]],
                    device: torch.device, src: int = 0) -> Dict[str, torch.Tensor]:
    """Ship a state dict from ``src`` to every rank as ONE flat fp32 buffer (a single
    broadcast: xGMI is point-to-point, so one large message beats hundreds of small ones).
    ``spec`` = [(key, shape, dtype)] must be known on every rank (it is a property of the
    architecture, see ``state_spec``); non-float entries travel as float and are cast back."""
    # no process group: nothing to ship.  A group of ONE rank still goes through the collective (on
    # a single-GPU box that is the only way the RCCL broadcast itself can be executed and tested).
    if not (dist.is_available() and dist.is_initialized()):
        assert state is not None
        return state
    total = sum(int(torch.Size(shape).numel()) for _, shape, _ in spec)
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        assert state is not None
        off = 0
        for key, shape, _ in spec:
            n = int(torch.Size(shape).numel())
            flat[off:off + n] = state[key].detach().reshape(-1).to(device=device, dtype=torch.float32)
            off += n
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    cpu = flat.cpu()
    for key, shape, dtype in spec:
        n = int(torch.Size(shape).numel())
        out[key] = cpu[off:off + n].reshape(shape).to(dtype).clone()
        off += n
    return out


def state_spec(state: Dict[str, torch.Tensor]) -> List[Tuple[str, Tuple[int, ...], torch.dtype]]:
    return [(k, tuple(v.shape), v.dtype) for k, v in state.items()]


def shard_streams(num_streams_total: int, rank: int, world: int) -> List[int]:
    """Stream ids owned by ``rank``: r, r+world, ... — no stream is split."""
    return list(range(rank, num_streams_total, world))


def shard_files_lpt(durations: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time assignment of whole files to ranks (one stream per GPU at a
    time, ``Benchmark``'s file loop ``inference.py:425-429`` partitioned instead of pooled)."""
    order = sorted(range(len(durations)), key=lambda i: -durations[i])
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += durations[i]
    return out


def gather_counts(values: Sequence[float], device: torch.device) -> List[List[float]]:
    """all_gather of a small per-rank vector (e.g. DER components); identity when world == 1."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [t.tolist()]
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    return [b.tolist() for b in bufs]


def timed_max_over_ranks(fn, device: Optional[torch.device] = None) -> float:
    """The timing bracket of ``bench.py``: barrier + device synchronise on both sides of ``fn()``,
    then the MAXIMUM elapsed time over the ranks (the job is as slow as its slowest rank).  Works
    without a process group (world 1) and without a GPU (``device`` None / CPU: gloo tests)."""
    import time
    multi = dist.is_available() and dist.is_initialized()      # also a group of one rank (see broadcast_state)
    on_gpu = device is not None and device.type == "cuda"

    def fence():
        if on_gpu:
            torch.cuda.synchronize(device)
        if multi:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)

    fence()
    t0 = time.perf_counter()
    fn()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if on_gpu else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def whole_job_rate(units_per_rank_per_step: int, steps: int, elapsed_max: float, world: int) -> float:
    """Units per second of the WHOLE job under weak scaling: every rank processed
    ``units_per_rank_per_step * steps`` units in (at most) ``elapsed_max`` seconds."""
    return world * units_per_rank_per_step * steps / elapsed_max
