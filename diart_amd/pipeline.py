"""Cross-stream driver of the hot path: N concurrent audio streams, one chunk each per step.

The reference batches consecutive windows of ONE stream (``inference.py:126-128``) and keeps one
pipeline object per stream (``blocks/diarization.py:121-125``); 64 concurrent real-time streams
(BASELINE.json config 2) therefore need a driver the reference does not have.  ``StreamBatch``
stacks the current window of every stream into one segmentation / embedding launch sequence and
steps N independent clustering states and (``tail=True``) N aggregation / binarisation states,
producing for each stream exactly what its own ``SpeakerDiarization.__call__`` computes at lines
186-232: segmentation, overlap-aware normalised embeddings, the permuted ``(frames,
max_speakers)`` scores and the speech turns of the region the step finalises.  ``AudioRing`` keeps
the rolling windows of host-fed streams on the device (only new samples are uploaded).

GPU schedule per step (``seg_split`` + ``emb_split`` HIP streams per lane; ``depth`` lanes so that
``depth`` consecutive steps can be in flight at once — the GPU halves of different steps are
independent, only the host-side clustering is sequential per stream):

    stream A_i : sub-batch i: dz_seg_forward (SincNet -> 4 x {x-projection GEMM, persistent
                 LSTM} -> MLP) -> dz_osp  --event-->        (the GEMMs of one sub-batch run
                 under the latency-bound recurrence of the other)
    stream B   : dz_emb_frames (SincNet -> 5 TDNN; independent of the segmentation, fills the
                 CUs the LSTM leaves idle)  <--wait all--  dz_emb_pool -> D2H (pinned)
    host       : clustering + output tail of step t-1 (C++ threads, fp64) while the GPU runs step t
"""
from __future__ import annotations

import ctypes as C
import os
import time as _time
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .blocks.aggregation import BatchedOutputTail
from .blocks.clustering import BatchedSpeakerClustering
from .features import Annotation, Segment
from .models import HipEmbedding, HipSegmentation, _as_rows

# lanes of a throughput engine (>= 64 streams per step on the matrix-core recurrence): profiles/r06*_lanes_grid.json
THROUGHPUT_LANES = 6


class AudioRing:
    """Device-resident rolling window of N streams (``dz_ring_*``): per step only the ``hop`` new
    samples of every stream are uploaded (32 KB instead of the 320 KB window the reference moves
    per chunk, ``blocks/segmentation.py:47``); ``StreamBatch.launch`` reads the window in place.
    The role of ``rearrange_audio_stream`` (reference ``operators.py:44-100``) for N streams."""

    def __init__(self, num_streams: int, window: int = 80000, hop: int = 8000, slack_blocks: int = 2,
                 device: Optional[torch.device] = None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.n, self.window, self.hop, self.slack = num_streams, window, hop, slack_blocks
        self._lib = _lib.load()
        self._h = _lib.vp()
        _lib.check(self._lib.dz_ring_create(_lib.context(self.device.index), num_streams, window, hop,
                                            slack_blocks, C.byref(self._h)), "dz_ring_create")
        self._keep: List[torch.Tensor] = []          # pinned blocks of in-flight copies
        self._readers: List[List[torch.cuda.Event]] = []   # per step: events after its forward passes

    def __del__(self):
        try:
            self._lib.dz_ring_destroy(self._h)
        except Exception:
            pass

    def reset(self):
        _lib.check(self._lib.dz_ring_reset(self._h), "dz_ring_reset")
        self._readers = []

    def push(self, block) -> bool:
        """``block``: (N, hop) float32 — a CPU tensor / ndarray (pinned memory makes the copy
        asynchronous) or a tensor on the ring's GPU.  Enqueued on the current HIP stream, after the
        forward passes of the window it is about to overwrite.  True once the window is complete."""
        if isinstance(block, np.ndarray):
            block = torch.from_numpy(block)
        assert block.dtype == torch.float32 and tuple(block.shape) == (self.n, self.hop), block.shape
        assert block.stride(1) == 1, "rows of the block must be contiguous"
        cur = torch.cuda.current_stream(self.device)
        # block t+1 overwrites samples of windows <= t - slack: wait for that window's readers
        while len(self._readers) > self.slack + 1:
            self._readers.pop(0)
        if len(self._readers) == self.slack + 1:
            for ev in self._readers[0]:
                cur.wait_event(ev)
        # Pinned host memory is mapped into the GPU's address space: the scatter kernel reads it in
        # place over PCIe (32 KB per stream per step), which keeps the upload off the SDMA queue —
        # there it can queue behind the D2H copy of a step whose results are not ready yet, and
        # the next step's forward passes then wait for the previous step's (measured: -17 %).
        # mode 2 = pinned host memory read IN PLACE by the scatter kernel (if the runtime can map it)
        mode = 1 if block.is_cuda else (2 if block.is_pinned() and _lib.exp_env("DZ_RING_ZERO_COPY", "1") != "0" else 0)
        _lib.check(self._lib.dz_ring_push(self._h, block.data_ptr(), block.stride(0), mode,
                                          cur.cuda_stream), "dz_ring_push")
        self._keep = (self._keep + [block])[-4:]
        return self.filled == self.window

    def _read_by(self, streams: List[torch.cuda.Stream]):
        """Called right after forward passes that read the current window were enqueued on
        ``streams``: records the ring's OWN events there (a caller's per-slot events are re-recorded
        when the slot is reused — waiting on such an event later means waiting for the NEWEST step
        that used the slot, which silently serialised consecutive steps)."""
        evs = []
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        self._readers.append(evs)

    # ---- streams that advance at their own pace (StreamServer) -------------------------------
    def push_rows(self, block: torch.Tensor, rows: Sequence[int]) -> None:
        """Row j of ``block`` (k, hop) is the next block of stream ``rows[j]`` (each row keeps its
        own write position).  Enqueued on the current HIP stream; a pinned host block is read in
        place by the GPU and must stay untouched until that stream has passed this point."""
        k = len(rows)
        assert block.dtype == torch.float32 and tuple(block.shape) == (k, self.hop) and block.stride(1) == 1
        mode = 1 if block.is_cuda else (2 if block.is_pinned() and _lib.exp_env("DZ_RING_ZERO_COPY", "1") != "0" else 0)
        arr = (C.c_int * k)(*[int(r) for r in rows])
        _lib.check(self._lib.dz_ring_push_rows(self._h, block.data_ptr(), block.stride(0), mode, arr, k,
                                               torch.cuda.current_stream(self.device).cuda_stream), "dz_ring_push_rows")
        self._keep = (self._keep + [block])[-4:]

    def filled_row(self, row: int) -> int:
        f = C.c_int()
        _lib.check(self._lib.dz_ring_filled_row(self._h, int(row), C.byref(f)), "dz_ring_filled_row")
        return int(f.value)

    def gather(self, rows: Sequence[int], out: torch.Tensor) -> torch.Tensor:
        """``out[j]`` <- the current window of stream ``rows[j]`` (device to device, current HIP
        stream); every listed stream must hold a complete window."""
        k = len(rows)
        assert out.is_cuda and out.dtype == torch.float32 and out.shape[0] >= k and out.shape[1] == self.window
        arr = (C.c_int * k)(*[int(r) for r in rows])
        _lib.check(self._lib.dz_ring_gather(self._h, arr, k, out.data_ptr(), out.stride(0),
                                            torch.cuda.current_stream(self.device).cuda_stream), "dz_ring_gather")
        return out[:k]

    def reset_row(self, row: int) -> None:
        _lib.check(self._lib.dz_ring_reset_row(self._h, int(row)), "dz_ring_reset_row")

    def raw(self) -> Tuple[int, int]:
        ptr, stride = _lib.vp(), C.c_longlong()
        _lib.check(self._lib.dz_ring_window(self._h, C.byref(ptr), C.byref(stride), None), "dz_ring_window")
        return int(ptr.value), int(stride.value)

    @property
    def filled(self) -> int:
        ptr, stride, f = _lib.vp(), C.c_longlong(), C.c_int()
        _lib.check(self._lib.dz_ring_window(self._h, C.byref(ptr), C.byref(stride), C.byref(f)), "dz_ring_window")
        return int(f.value)

    def snapshot(self) -> torch.Tensor:
        """Contiguous copy (N, window) of the current window (what ``rearrange_audio_stream`` would
        emit); the forward passes do not need it, they read the ring in place."""
        out = torch.empty((self.n, self.window), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.dz_ring_read(self._h, out.data_ptr(),
                                          torch.cuda.current_stream(self.device).cuda_stream), "dz_ring_read")
        return out


class StreamBatch:
    def __init__(self, segmentation: HipSegmentation, embedding: HipEmbedding, num_streams: int,
                 tau_active: float = 0.6, rho_update: float = 0.3, delta_new: float = 1.0,
                 gamma: float = 3, beta: float = 10, max_speakers: int = 20,
                 normalize_embedding_weights: bool = False,
                 device: Optional[torch.device] = None, cluster_threads: int = 8,
                 seg_split: Optional[int] = None, emb_split: Optional[int] = None,
                 tail: bool = False, duration: float = 5.0, step: float = 0.5,
                 latency: Optional[float] = None, depth: Optional[int] = None, *, lanes: Optional[int] = None,
                 recurrence: Optional[str] = None, inflight: Optional[int] = None, wait: Optional[str] = None,
                 warmup: Optional[int] = None, serial: bool = False):
        """Engine parameters (``DZ_ENGINE`` overrides them, config.py):
        ``lanes`` (= ``depth``): steps the GPU works on at once, each with its own HIP streams, handles and scratch
        arenas (~0.85 GB per lane at 64 streams) — default 2, or ``THROUGHPUT_LANES`` for a throughput engine;
        ``recurrence``: the LSTM recurrence kernel, "valu" | "0" | "3" | "4" — default: the matrix-core kernel of
        ``weights.THROUGHPUT_LSTM_VARIANT`` for a throughput engine (>= 64 streams per step, default precision),
        else the model's own;
        ``inflight``: steps a throughput caller keeps between ``launch`` and ``finish`` (default lanes + 1; lanes + 2 for a
        throughput engine);
        ``wait``: how the host waits for a step, "spin" | "block" | "auto" (by the cores this rank has);
        ``warmup``: warm steps on silence before the first real step of a window size (default 10, 0 = off);
        ``serial``: MEASUREMENT engine — one lane whose segmentation and embedding chains share ONE HIP stream, so
        that no two kernels ever overlap and a kernel's bracketed duration is its alone-time (what
        ``rocprofv3 --kernel-trace --stats`` of the same run reports): ``bench.py``'s roofline pass."""
        from .config import setting
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.seg, self.emb = segmentation.to(self.device), embedding.to(self.device)
        self.device = self.seg.device
        self.n = num_streams
        self.gamma, self.beta, self.norm_w = float(gamma), float(beta), bool(normalize_embedding_weights)
        self.clustering = BatchedSpeakerClustering(num_streams, tau_active, rho_update, delta_new,
                                                   max_speakers, cluster_threads)
        self.max_speakers = max_speakers
        # optional output tail (DelayedAggregation + Binarize of every stream, C++): built on the
        # first step, when the number of frames per chunk is known
        self.with_tail, self.duration, self.step = bool(tail), float(duration), float(step)
        self.latency = self.step if latency is None else float(latency)
        self.tau_active, self.cluster_threads = tau_active, cluster_threads
        self.tail: Optional[BatchedOutputTail] = None
        self._t = 0                                        # launches so far (lane selection)
        self._steps = np.zeros(num_streams, dtype=np.int64)   # windows seen by each stream slot
        # sub-batches per network, each on its own HIP stream with its own scratch arena: the
        # x-projection GEMM of one sub-batch runs under the latency-bound recurrence of another
        self.seg_split = max(1, min(int(_lib.exp_env("DZ_SEG_SPLIT", "1") if seg_split is None else seg_split), num_streams))
        self.emb_split = max(1, min(int(_lib.exp_env("DZ_EMB_SPLIT", "1") if emb_split is None else emb_split), num_streams))
        # HIP stream priorities (0 normal, -1 high).  The segmentation chain is the long dependent one
        # (4 recurrences + their projections: ~2.2 ms in the pipeline, two lanes): its streams get the
        # high priority — round 3, two same-visit pairs: 1.215 vs 1.233 and 1.159 vs 1.180 ms per step
        # (gpurun_out/visit_r3l.log; in round 2, with longer small kernels in that chain, the effect was
        # inside the noise).  The embedding chain waits for the segmentation anyway and stays normal.
        pa, pb = int(_lib.exp_env("DZ_PRIO_A", "-1")), int(_lib.exp_env("DZ_PRIO_B", "0"))
        # `depth` lanes, each with its own HIP streams and scratch arenas: step t runs on lane
        # t % depth, so a caller that keeps `depth` tickets between launch() and finish() has that
        # many steps on the GPU at once (the latency-bound recurrence of one step under the GEMMs
        # of the others).  A lane is reused in stream order, which also orders its arenas.
        # Round 5: with >= 64 streams per step the recurrence runs on the matrix cores (16 chains per workgroup: a
        # seventh of the CU-time, twice the latency of a layer) and SIX steps are in flight instead of two — the longer
        # dependent chain of a lane hides under five other steps, and the 128 CUs the one-chain-per-CU recurrence held
        # for most of a step go to the GEMM-shaped kernels: 30 300 -> 33 500 xRT in same-visit pairs
        # (profiles/r05u_recurrence_lanes_grid.json; in round 4, with the slower front end, the same pair measured
        # equal).  Fewer streams (FileBatch's 32 windows per step lost 11 % with it), the exact-f32 precision and an
        # explicit DZ_LSTM keep what they had; the synchronous blocks API always runs the low-latency recurrence.
        rec = setting("recurrence", recurrence, None)
        split = getattr(self.seg, "precision", "f32") == "f16x3"
        self.throughput = num_streams >= 64 and rec is None and split and getattr(self.seg, "recurrence", "valu") == "valu"
        # the kernel this engine's handles run: None = the model's own struct
        self.recurrence = self.seg.throughput_recurrence() if self.throughput else (str(rec) if rec is not None and split else None)
        if depth is not None and lanes is not None and int(depth) != int(lanes):
            raise ValueError(f"StreamBatch: depth={depth} and lanes={lanes} name the same thing")
        matrix_core = self.recurrence not in (None, "valu")
        self.depth = max(1, int(setting("lanes", lanes if lanes is not None else depth,
                                        THROUGHPUT_LANES if (self.throughput or (matrix_core and num_streams >= 64)) else 2, int)))
        # How many launched-but-unfinished steps a throughput caller (bench.py, FileBatch) keeps: `depth` lanes
        # run concurrently, the steps beyond that wait IN THE STREAMS of their lane, so that a lane's next
        # step starts the moment the previous one ends instead of after the host has come back from
        # finish() (clustering of an older step + launch overhead: ~0.5 ms per step, during which the
        # lane's segmentation stream sat empty).
        # Round 6: a throughput engine keeps lanes + 2 (the second spare ticket covers the host's own launch + tail time of
        # a step: +3 % in the 20-step form, profiles/r06l_inflight_grid.json); the two-lane engines keep lanes + 1.
        spare = 2 if (self.throughput or (matrix_core and num_streams >= 64)) else 1
        self.max_inflight = max(self.depth, int(setting("inflight", inflight, self.depth + spare, int)))
        self.warmup_steps = max(0, int(setting("warmup", warmup, 10, int)))
        # DZ_SHARED_EMB=1: ONE set of embedding streams serves every lane in step order and the
        # pooling of step t is enqueued `lag` = depth - 1 launches later, behind the frame features
        # of the following steps (only the last two kernels of the embedding network wait for the
        # segmentation; by then it has finished, so the stream never blocks on it).  That is what a
        # latency-heavy segmentation needs (the matrix-core recurrence, DZ_LSTM=0..3, with depth >= 3:
        # fewer streams than lanes x 2, the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
        # queues).  With the default recurrence kernel two full lanes measured 8 % faster (24.7 k vs
        # 22.8 k xRT), so the default is one embedding stream per lane.
        # DZ_ABLATE=noemb | noseg: TIMING EXPERIMENT (results are wrong): one of the two networks is not launched,
        # to see what the step costs when the other one has the chip to itself
        self._ablate = _lib.exp_env("DZ_ABLATE", "")
        # How the host waits.  With cores to spare the launching thread spins on the step's `done` event (lowest
        # latency) and the pool's workers poll 40 us for the next job; a rank that has ~2 cores for itself (8 ranks
        # on a node's 16 usable cores) cannot afford either: the event is then a blocking one (the thread sleeps in
        # the driver until the GPU signals) and the workers sleep at once.  DZ_WAIT=spin | block, DZ_POOL_SPIN_US.
        from .hostinfo import usable_cores
        try:
            ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
        except ValueError:
            ranks_here = 1
        self.cores_per_rank = usable_cores() / ranks_here
        mode = str(setting("wait", wait, "auto"))
        if mode not in ("auto", "spin", "block"):
            raise ValueError(f"StreamBatch: wait={mode!r} (expected auto | spin | block)")
        self.blocking_wait = mode == "block" or (mode == "auto" and self.cores_per_rank < 4)
        # idle pool workers poll 40 us for the next job when this rank has cores to spare, else they sleep at once
        # (process-wide setting: the newest engine's situation decides)
        _lib.load().dz_host_pool_set_spin(0 if self.cores_per_rank < 4 else 40)
        self.shared_stats = _lib.exp_env("DZ_SHARED_STATS", "1") != "0"
        self.shared_emb = _lib.exp_env("DZ_SHARED_EMB", "0") != "0"
        self.lag = self.depth - 1 if self.shared_emb else 0
        # DZ_SEG_FRONT=1: the stateless front half of the segmentation network (SincNet + the first
        # x-projection, dz_seg_front) gets a stream of its own per lane.  launch(t + depth) is called
        # before finish(t), so on one stream the front half of step t + depth queues BEHIND the
        # recurrences of step t; on its own stream it runs under them and the lane's dependent chain is
        # the back half only (dz_seg_back: 4 recurrences, 3 projections, the MLP head).
        self.seg_front = _lib.exp_env("DZ_SEG_FRONT", "0") != "0"
        pf = int(_lib.exp_env("DZ_PRIO_F", "0"))
        mk = lambda prio, k: [torch.cuda.Stream(self.device, priority=prio) for _ in range(k)]
        shared_b = mk(pb, self.emb_split) if self.shared_emb else None
        self.lanes = [dict(a=mk(pa, self.seg_split), b=shared_b or mk(pb, self.emb_split),
                           f=mk(pf, self.seg_split) if self.seg_front else None)
                      for _ in range(self.depth)]
        self.serial = bool(serial)
        if self.serial:
            if self.depth != 1 or self.seg_split != 1 or self.emb_split != 1 or self.seg_front:
                raise ValueError("StreamBatch(serial=True) is one lane with one stream: lanes=1, no sub-batches")
            self.lanes[0]["b"] = self.lanes[0]["a"]
        self.streams_a, self.streams_b = self.lanes[0]["a"], self.lanes[0]["b"]
        self.stream_a, self.stream_b = self.streams_a[0], self.streams_b[0]
        self.num_hip_streams = self.depth * self.seg_split + self.emb_split * (1 if self.shared_emb else self.depth)
        # DZ_CONV0_PAIR=1 (experiments build): the first SincNet stage of BOTH networks in one launch
        # (dz_sinc_conv0_pair: one split of the normalised window, 160 filters, one parking of the samples) on the
        # lane's first segmentation stream, the embedding stream waits for it.  Correct, and slower: 1.23 vs 1.15 ms
        # per step (csrc/k_front.hip has the numbers).
        self.conv0_pair = (_lib.exp_env("DZ_CONV0_PAIR", "0") == "1" and self.shared_stats
                           and getattr(self.seg, "precision", "") == "f16x3" and getattr(self.emb, "precision", "") == "f16x3"
                           and self.seg_split == 1 and self.emb_split == 1 and not self.seg_front and not self._ablate
                           and hasattr(self.emb, "_state") and type(self.emb).__name__ == "HipEmbedding")
        self._pair = None                                  # PackedConv0Pair, built at the first launch
        self._pending: List[dict] = []                     # launched, pooling not enqueued yet
        # where finish() spends its time: waiting for the GPU vs clustering + output tail on the host
        self.host_seconds = {"wait": 0.0, "work": 0.0}
        # measurement (bench.py sets it to a list): launches that took the host more than 2 ms, by the call they sat in
        self.slow_launches: Optional[list] = None
        self.d2h_by_kernel = True          # (False: the hipMemcpyAsync pair, kept for the A/B of profiles/r06z_launch_stalls.json)
        self._sub: dict = {}
        self._warmed: set = set()
        self._warming, self._real_steps = False, 0       # (see _warm_up)
        self._real_launches = 0                          # launches issued for a caller (not by _warm_up)
        self._slots: List[dict] = []
        self._lib = _lib.load()
        self._ctx = _lib.context(self.device.index)

    def set_host_threads(self, n: int) -> int:
        """Host threads of the per-stream CPU stages (clustering, aggregation + binarisation) from now on."""
        n = max(1, int(n))
        self.cluster_threads = self.clustering.num_threads = n
        if self.tail is not None:
            self.tail.num_threads = n
        return n

    def reset(self, slot: Optional[int] = None):
        """Forget the clustering / aggregation state of every stream, or of one slot."""
        self.clustering.reset(slot)
        if self.tail is not None:
            self.tail.reset(slot)
        if slot is None:
            self._steps[:] = 0
        else:
            self._steps[slot] = 0

    # ------------------------------------------------------------------ GPU half
    def _slot(self, F: int, K: int, D: int) -> dict:
        for s in self._slots:
            if not s["busy"] and s["shape"] == (F, K, D):
                return s
        return self._new_slot(F, K, D)

    def _new_slot(self, F: int, K: int, D: int) -> dict:
        n, dev = self.n, self.device
        s = dict(shape=(F, K, D), busy=False,
                 seg=torch.empty((n, F, K), dtype=torch.float32, device=dev),
                 w=torch.empty((n, K, F), dtype=torch.float32, device=dev),
                 emb=torch.empty((n, K, D), dtype=torch.float32, device=dev),
                 stats=torch.empty((n, self._lib.dz_wave_stats_floats()), dtype=torch.float32, device=dev),
                 seg_h=torch.empty((n, F, K), dtype=torch.float32).pin_memory(),
                 emb_h=torch.empty((n, K, D), dtype=torch.float32).pin_memory(),
                 ev_seg=[torch.cuda.Event() for _ in range(self.seg_split)],
                 ev_front=[torch.cuda.Event() for _ in range(self.seg_split)],
                 ev_emb=[torch.cuda.Event() for _ in range(self.emb_split - 1)],
                 ev_frames=[torch.cuda.Event() for _ in range(self.emb_split)],
                 ev_in=torch.cuda.Event(), ev_conv0=torch.cuda.Event(),
                 done=torch.cuda.Event(blocking=self.blocking_wait))
        self._slots.append(s)
        return s

    @staticmethod
    def _ranges(n: int, parts: int) -> List[Tuple[int, int]]:
        return [(n * i // parts, n * (i + 1) // parts) for i in range(parts)]

    def _handles(self, S: int, lane: int):
        """C handles (one scratch arena each, owned by this object) of the sub-batches of one lane
        for windows of S samples."""
        got = self._sub.get((S, lane))
        if got is None:
            sa, sb = self._ranges(self.n, self.seg_split), self._ranges(self.n, self.emb_split)
            hs = [self.seg._create(S, max(1, -(-self.n // self.seg_split)), throughput=self.throughput,
                                   recurrence=None if self.throughput else self.recurrence) for _ in sa]
            he = [self.emb._create(S, max(1, -(-self.n // self.emb_split))) for _ in sb]
            got = self._sub[(S, lane)] = (hs, he, sa, sb)
        return got

    def __del__(self):
        try:
            for hs, he, _, _ in self._sub.values():
                for h in hs:
                    self.seg._destroy(h)
                for h in he:
                    self.emb._destroy(h)
        except Exception:
            pass

    def launch(self, waves, starts=None, slots: Optional[Sequence[int]] = None) -> dict:
        """Enqueue the GPU work for one step.  ``waves``: (N, S) or (N, 1, S) float32 on the GPU
        (a strided rolling-window view is used in place) or an ``AudioRing`` holding a complete
        window.  ``starts``: start time in seconds of each
        stream's window (default: the number of windows that stream slot has seen since its last
        ``reset`` x ``step``), used by the output tail only.
        ``slots``: which of the ``num_streams`` clustering / tail states the rows belong to, for a
        step in which only some streams have a new window (``len(slots)`` rows; default: all, in
        order).  Returns a ticket for ``finish``."""
        ring = waves if isinstance(waves, AudioRing) else None
        if ring is not None:
            assert ring.filled == ring.window, "the ring does not hold a complete window yet"
            rows, N, S = None, ring.n, ring.window
            base, stride = ring.raw()
        else:
            rows = _as_rows(waves)
            N, S = rows.shape
            base, stride = rows.data_ptr(), (rows.stride(0) if N > 1 else S)
        if slots is None:
            assert N == self.n, f"expected {self.n} streams, got {N}"
        else:
            slots = [int(i) for i in slots]
            assert N == len(slots) and 1 <= N <= self.n and len(set(slots)) == N, "bad slots"
            assert all(0 <= i < self.n for i in slots), "slot out of range"
        slot = self._launch_rows(base, stride, N, S, rows, ring)
        slot["slots"] = slots
        idx = np.arange(self.n) if slots is None else np.asarray(slots, dtype=np.int64)
        slot["starts"] = self._steps[idx] * self.step if starts is None else starts
        self._steps[idx] += 1
        return slot

    def _launch_rows(self, base: int, stride: int, N: int, S: int, keep=None, ring: Optional[AudioRing] = None) -> dict:
        """The GPU half of a step for N <= num_streams rows of S samples at ``base + i * stride``
        floats: segmentation (+ OSP weights) and frame features on this step's lane, pooling behind
        them.  Knows nothing about which stream / file a row belongs to (``launch`` and ``FileBatch``
        decide that)."""
        assert 1 <= N <= self.n
        F, K, D = self.seg.num_frames(S), None, self.emb.dimension
        if (S, 0) not in self._sub:
            # The scratch arenas of EVERY lane before the first kernel of this window size is enqueued
            # (~0.85 GB of hipMalloc + hipMemset per lane): no device allocation while kernels run.  (This
            # did NOT remove the 21 - 39 ms start-up stall some runs show in their third step — it persisted
            # with the arenas allocated up front; profiles/README.md)
            for ln in range(self.depth):
                self._handles(S, ln)
            # ... and the output tail's host state NOW, while the GPU is idle: built lazily in the first
            # finish() — 64 aggregation states (30 MB of fresh host memory, mapped and first touched) + 7.5 MB
            # of result buffers while two steps were on the GPU — it froze every resident kernel for 20 - 40 ms
            # (tools/stall_probe.py: gone when only this is moved; the first clustering call, which starts the
            # worker threads, is not the trigger).  This was the start-up stall of rounds 1 - 3.
            if self.with_tail and self.tail is None:
                self.tail = BatchedOutputTail(self.n, F, self.max_speakers, self.step, self.latency,
                                              self.tau_active, num_threads=self.cluster_threads)
        if S not in self._warmed:
            self._warmed.add(S)
            self._warm_up(S)
        if not self._warming:
            self._real_launches += 1
        lane = self.lanes[self._t % self.depth]
        hsegs, hembs, _, _ = self._handles(S, self._t % self.depth)
        marks = [("start", _time.perf_counter())] if self.slow_launches is not None else None
        # sub-batch ranges of THIS step's rows (each at most the capacity its handle was built for)
        sa, sb = self._ranges(N, self.seg_split), self._ranges(N, self.emb_split)
        K = self.seg.num_speakers
        if not any(s["shape"] == (F, K, D) for s in self._slots):
            # every in-flight slot up front: a pinned-memory allocation made while kernels are
            # running stalls the queues for tens of milliseconds (seen as one 40 ms "kernel" in the
            # rocprofv3 trace of the second step)
            for _ in range(self.max_inflight + self.lag):
                self._new_slot(F, K, D)
        slot = self._slot(F, K, D)
        slot["busy"] = True
        lib, esz = self._lib, 4
        cur = torch.cuda.current_stream(self.device)
        slot["ev_in"].record(cur)                       # inputs produced on the caller's stream
        # InstanceNorm1d(1) statistics of the windows ONCE for both networks (each SincNet would
        # otherwise make its own pass over the same 20 MB).  They go first on the lane's first
        # segmentation stream — the high-priority one: on the caller's (normal-priority) stream this 14 us
        # kernel sat 60 - 130 us in the queue in front of BOTH chains — and every other stream of the
        # step waits for `ev_in` re-recorded behind it.
        stats = slot["stats"] if self.shared_stats else None
        front = lane["f"]                               # None: both halves on the `a` streams
        if stats is not None:
            a0 = (front or lane["a"])[0]
            a0.wait_event(slot["ev_in"])
            _lib.check(lib.dz_wave_stats(self._ctx, base, stride, N, S, stats.data_ptr(), a0.cuda_stream),
                       "dz_wave_stats")
            slot["ev_in"].record(a0)
            if marks is not None:
                marks.append(("wave_stats", _time.perf_counter()))
        pair = self.conv0_pair and stats is not None
        if pair:
            if self._pair is None:
                from .weights import PackedConv0Pair
                self._pair = PackedConv0Pair(self.seg._state, self.emb._state, self.device)
            a0 = lane["a"][0]
            _lib.check(lib.dz_sinc_conv0_pair(hsegs[0], hembs[0], base, stride, N, stats.data_ptr(),
                                              self._pair.planes.data_ptr(), self._pair.bsum.data_ptr(), a0.cuda_stream),
                       "dz_sinc_conv0_pair")
            slot["ev_conv0"].record(a0)
        for j, ((i0, i1), h, a, ev) in enumerate(zip(sa, hsegs, lane["a"], slot["ev_seg"])):
            a.wait_event(slot["ev_in"])
            if i1 == i0:                                 # fewer rows than sub-batches
                ev.record(a)
                continue
            if stats is not None:
                _lib.check(lib.dz_seg_use_wave_stats(h, stats[i0:].data_ptr()), "dz_seg_use_wave_stats")
            if front is not None:
                f = front[j]
                f.wait_event(slot["ev_in"])
                _lib.check(lib.dz_seg_front(h, base + i0 * stride * esz, stride, i1 - i0, f.cuda_stream),
                           "dz_seg_front")
                slot["ev_front"][j].record(f)
                a.wait_event(slot["ev_front"][j])
                _lib.check(lib.dz_seg_back(h, i1 - i0, slot["seg"][i0:i1].data_ptr(), self.gamma, self.beta,
                                           int(self.norm_w), slot["w"][i0:i1].data_ptr(), a.cuda_stream),
                           "dz_seg_back")
                ev.record(a)
                continue
            if self._ablate == "noseg":
                ev.record(a)
                continue
            # segmentation + the OSP weights of its output (one launch sequence, no dz_osp of its own)
            _lib.check(lib.dz_seg_forward_osp(h, base + i0 * stride * esz, stride, i1 - i0,
                                              slot["seg"][i0:i1].data_ptr(), self.gamma, self.beta,
                                              int(self.norm_w), slot["w"][i0:i1].data_ptr(), a.cuda_stream),
                       "dz_seg_forward_osp")
            ev.record(a)
            if marks is not None:
                marks.append(("seg_forward_osp", _time.perf_counter()))
        for (i0, i1), h, b, ev in zip(sb, hembs, lane["b"], slot["ev_frames"]):
            b.wait_event(slot["ev_conv0"] if pair else slot["ev_in"])
            if i1 > i0 and self._ablate != "noemb":
                if stats is not None:
                    _lib.check(lib.dz_emb_use_wave_stats(h, stats[i0:].data_ptr()), "dz_emb_use_wave_stats")
                _lib.check(lib.dz_emb_frames(h, base + i0 * stride * esz, stride, i1 - i0, b.cuda_stream),
                           "dz_emb_frames")
            ev.record(b)
            if marks is not None:
                marks.append(("emb_frames", _time.perf_counter()))
        slot["rows"], slot["slots"] = N, None
        slot["keep"] = keep                              # keep the view alive until the GPU is done
        slot["pool"] = (lane, hembs, sa, sb, N, K, F)     # what _enqueue_pool needs
        if ring is not None:                             # pushes `slack` steps from now wait for these
            ring._read_by(list(lane["a"]) + list(lane["b"]) + list(front or []))
        self._pending.append(slot)
        while len(self._pending) > self.lag:
            self._enqueue_pool(self._pending.pop(0))
        if marks is not None:
            marks.append(("pool", _time.perf_counter()))
            if marks[-1][1] - marks[0][1] > 2e-3:           # a launch that took more than 2 ms: where
                self.slow_launches.append([self._t] + [(b[0], round(1e3 * (b[1] - a[1]), 2)) for a, b in zip(marks, marks[1:])])
        self._t += 1
        return slot

    def _warm_up(self, S: int) -> None:
        """A few overlapped steps on silence before the first real one of this window size.  Besides the
        start-up stall proper (the output tail's host buffers, now allocated before the first launch: see
        `_launch_rows`), a fresh process shows two 7 - 12 ms intervals within its first ~8 steps
        (tools/stall_probe.py, no tracer; profiles/r03_f_stall_probe.txt): one-time set-up for the first
        launches of each kernel on each queue — gone after warm steps, while warming the streams, the copy
        path and the events alone did not move them.  Complete steps (GPU + host half: the worker pool's
        threads start here too) while no stream has any state yet, GPU-only steps for a second window size
        later on; ~0.1 s of construction time instead of latency spikes in a live stream.  ``warmup=<steps>``
        (0 = off)."""
        steps = self.warmup_steps
        if steps <= 0:
            return
        zeros = torch.zeros((self.n, S), dtype=torch.float32, device=self.device)
        saved = dict(self.host_seconds)
        inflight: List[dict] = []
        # Complete steps (and the reset() that forgets them) only while NO launch has been issued for a caller:
        # counting finished steps instead (round 3) let a second window size, or a first finish() that comes
        # after several launches (FileBatch never calls finish), wipe the clustering / tail state and the
        # step counters under tickets that were still in flight.
        if self._real_launches == 0:                     # complete steps, then forget them
            self._warming = True
            try:
                for _ in range(steps):
                    inflight.append(self.launch(zeros))
                    if len(inflight) >= self.max_inflight:
                        self.finish(inflight.pop(0))
                while inflight:
                    self.finish(inflight.pop(0))
            finally:
                self._warming = False
            torch.cuda.synchronize(self.device)
            self.reset()
            self._t = 0
        else:                                            # GPU-only steps: no stream state is touched
            base, stride = zeros.data_ptr(), zeros.stride(0)
            t_saved = self._t
            self._warming = True
            try:
                for _ in range(steps):
                    inflight.append(self._launch_rows(base, stride, self.n, S, zeros))
                    if len(inflight) >= self.max_inflight:
                        t = inflight.pop(0)
                        self._wait(t)
                        t["busy"], t["keep"] = False, None
                for t in inflight:
                    self._wait(t)
                    t["busy"], t["keep"] = False, None
            finally:
                self._warming = False
                self._t = t_saved
            torch.cuda.synchronize(self.device)
        self.host_seconds.update(saved)

    def _enqueue_pool(self, slot: dict):
        """Statistics pooling + Linear + normalisation of a launched step (they consume the OSP
        weights, i.e. wait for its segmentation), the copy of its results to pinned memory and its
        `done` event, on the embedding stream(s)."""
        lane, hembs, sa, sb, N, K, F = slot["pool"]
        lib = self._lib
        for (i0, i1), h, b in zip(sb, hembs, lane["b"]):
            if i1 == i0:
                continue
            for (j0, j1), ev in zip(sa, slot["ev_seg"]):
                if j0 < i1 and i0 < j1:
                    b.wait_event(ev)
            if self._ablate == "noemb":
                if not slot.get("_filled"):       # something the clustering accepts
                    with torch.cuda.stream(b):
                        slot["emb"].fill_(0.04)
                    slot["_filled"] = True
                continue
            _lib.check(lib.dz_emb_pool(h, slot["w"][i0:i1].data_ptr(), i1 - i0, K, F, 1,
                                       slot["emb"][i0:i1].data_ptr(), b.cuda_stream), "dz_emb_pool")
        tr = self.slow_launches is not None
        m0 = _time.perf_counter() if tr else 0.0
        b0 = lane["b"][0]
        for b, ev in zip(lane["b"][1:], slot["ev_emb"]):
            ev.record(b)
            b0.wait_event(ev)
        # Results -> pinned memory by a kernel of our own.  The two hipMemcpyAsync (tensor.copy_) that stood here
        # blocked the launching thread for a whole step's latency about once per 150 steps (csrc/ring.hip).
        D = slot["emb"].shape[2]
        if self.d2h_by_kernel:
            _lib.check(lib.dz_results_to_host(self._ctx, slot["seg"].data_ptr(), slot["seg_h"].data_ptr(), N * F * K,
                                              slot["emb"].data_ptr(), slot["emb_h"].data_ptr(), N * K * D,
                                              b0.cuda_stream), "dz_results_to_host")
        else:
            with torch.cuda.stream(b0):
                slot["seg_h"][:N].copy_(slot["seg"][:N], non_blocking=True)
                slot["emb_h"][:N].copy_(slot["emb"][:N], non_blocking=True)
        m2 = _time.perf_counter() if tr else 0.0
        slot["done"].record(b0)
        if tr and _time.perf_counter() - m0 > 2e-3:
            self.slow_launches.append(["pool tail", ("results to host", round(1e3 * (m2 - m0), 2)),
                                       ("record done", round(1e3 * (_time.perf_counter() - m2), 2))])
        slot["pool"] = None

    # ------------------------------------------------------------------ host half
    def finish(self, ticket: dict, want_scores: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """Wait for the step's GPU work, run the N clustering updates.
        -> (segmentation (N,F,K) f32, embeddings (N,K,D) f32, scores (N,F,G) f64 | None, assign (N,K)).
        The arrays (and ``ticket["tail"]``) are views of buffers that a later ``launch`` / ``finish``
        reuses: copy what has to outlive the next step."""
        import time as _time
        self._wait(ticket)
        if not self._warming:
            self._real_steps += 1
        t1 = _time.perf_counter()
        # The ticket's slot is ALWAYS handed back: an exception between here and the end
        # used to leak it, and every later launch then allocated a new pinned slot while kernels were
        # running.  The range flag (an f16x3 operand beyond +-65504 is an error, not a clamp) is
        # checked first: a flagged step's results are clamped garbage and must not reach the
        # clustering state of the streams; the caller gets error 6 for THIS step and the streams
        # skip one window.  The flag is one word per device: with `depth` steps in flight it may have
        # been raised by a kernel of a later step, whose own finish() then passes — over-reporting by
        # at most `depth - 1` steps, never under-reporting.
        try:
            _lib.range_check(self.device.index)
            N, slots = ticket["rows"], ticket["slots"]
            seg = ticket["seg_h"].numpy()[:N]
            emb = ticket["emb_h"].numpy()[:N]
            scores, assign = self.clustering(seg, emb, want_scores or self.with_tail, slots=slots)
            if self.with_tail:
                # diarization.py:190,203-232 for every stream: aggregate the overlapping windows of the
                # region [t - latency, t - latency + step) and binarise it
                if self.tail is None:
                    self.tail = BatchedOutputTail(self.n, seg.shape[1], self.max_speakers, self.step,
                                                  self.latency, self.tau_active,
                                                  num_threads=self.cluster_threads)
                ticket["tail"] = self.tail(scores, ticket["starts"], self.duration / seg.shape[1], slots=slots)
        finally:
            ticket["busy"] = False
            ticket["keep"] = None
            self.host_seconds["work"] += _time.perf_counter() - t1
        return seg, emb, scores, assign

    def _wait(self, ticket: dict) -> None:
        """Block until the GPU half of the step is done and its results are in pinned host memory."""
        import time as _time
        t0 = _time.perf_counter()
        while ticket["pool"] is not None:               # its pooling is still held back: flush in order
            self._enqueue_pool(self._pending.pop(0))
        ticket["done"].synchronize()
        self.host_seconds["wait"] += _time.perf_counter() - t0

    def __call__(self, waves: torch.Tensor):
        return self.finish(self.launch(waves))

    def diarize(self, waves: torch.Tensor, starts=None):
        """One step of every stream, end to end: -> list of N ``Annotation`` (the speech turns of the
        region this step finalises, what ``SpeakerDiarization.__call__`` returns per chunk) —
        requires ``tail=True``."""
        assert self.with_tail, "construct StreamBatch(tail=True)"
        ticket = self.launch(waves, starts)
        self.finish(ticket, want_scores=False)
        _, _, _, _, turns, nturns = ticket["tail"]
        return [BatchedOutputTail.annotation(turns[i], int(nturns[i])) for i in range(self.n)]


class FileBatch:
    """Several FILES of one rank through the hot path at once — the file-parallel evaluation
    (BASELINE.json configs 1 / 4) at the batcher's rate.

    The reference's ``Benchmark`` (``inference.py:392-432``) feeds one file at a time: batches of 32
    consecutive windows go through ``SpeakerDiarization.__call__``, whose per-chunk Python loop
    (``blocks/diarization.py:193-232``) then runs clustering, aggregation and binarisation.  Only
    that loop is sequential, and only within a file.  Here one GPU step carries ``rows`` windows:
    the next ``rows // k`` CONSECUTIVE windows of each of the ``k`` files that are open (file-major
    rows, copied device to device from the file's resident audio); the host half hands every file
    to a thread that walks its windows in order through that file's own clustering / aggregation
    state (``dz_file_step_batch``), ``depth`` GPU steps in flight.  A file that ends frees its slot
    for the next one.  Per file the speech turns — hence the RTTM — are those of the
    one-file-at-a-time path (``tests/test_gpu_der.py``).

    ``run(files)``: ``files`` = iterable of ``(uri, padded waveform float32 (samples,), shift)``;
    the waveform already carries ``config.get_padding`` and a zero-filled last block (what
    ``file_blocks`` emits), ``shift`` is the pipeline's ``timestamp_shift``.  Returns
    ``{uri: Annotation}`` (``PredictionAccumulator`` semantics: every chunk's turns, then
    ``support(patch_collar)``)."""

    def __init__(self, segmentation: HipSegmentation, embedding: HipEmbedding, *, rows: int = 64,
                 max_files: int = 16, tau_active: float = 0.6, rho_update: float = 0.3, delta_new: float = 1.0,
                 gamma: float = 3, beta: float = 10, max_speakers: int = 20,
                 normalize_embedding_weights: bool = False, duration: float = 5.0, step: float = 0.5,
                 latency: Optional[float] = None, sample_rate: int = 16000,
                 device: Optional[torch.device] = None, threads: int = 8, patch_collar: float = 0.05,
                 recurrence: Optional[str] = "valu", lanes: Optional[int] = 2):
        """``recurrence`` / ``lanes``: the engine under the files.  A file job is short and its files end at different
        times: the latency-form recurrence on two lanes finishes 16 ten-minute files 13 - 40 % sooner than the
        throughput engine a 64-row ``StreamBatch`` would pick for itself (profiles/r06q_file_batch_engine.json:
        25 800 - 26 700 vs 17 900 - 23 600 chunks/s); None = let ``StreamBatch`` choose."""
        self.rows, self.max_files = int(rows), max(1, min(int(max_files), int(rows)))
        self.engine = StreamBatch(segmentation, embedding, self.rows, tau_active, rho_update, delta_new, gamma,
                                  beta, max_speakers, normalize_embedding_weights, device=device,
                                  cluster_threads=threads, tail=False, duration=duration, step=step,
                                  latency=latency, recurrence=recurrence, lanes=lanes)
        self.device = self.engine.device
        self.duration, self.step, self.sr = float(duration), float(step), int(sample_rate)
        self.latency = self.step if latency is None else float(latency)
        self.S, self.hop = int(round(sample_rate * duration)), int(round(sample_rate * step))
        assert self.hop % 4 == 0, "the step must be a multiple of 4 samples (16-byte aligned windows)"
        self.tau, self.rho, self.delta, self.G = tau_active, rho_update, delta_new, int(max_speakers)
        self.threads, self.patch_collar = int(threads), patch_collar
        self._lib = _lib.load()
        self._clu: List = []          # per file slot: dz_clu handle
        self._tails: Optional[BatchedOutputTail] = None
        self._stage: List[torch.Tensor] = []
        self._pool, self._lent = FileBatch._PinnedPool(), {}
        self.chunks_done = 0

    # ------------------------------------------------------------------ per-slot state
    def _ensure_state(self, F: int) -> None:
        if self._tails is not None:
            return
        for _ in range(self.max_files):
            h = _lib.vp()
            _lib.check(self._lib.dz_clu_create(self.tau, self.rho, self.delta, self.G, C.byref(h)), "dz_clu_create")
            self._clu.append(h)
        self._tails = BatchedOutputTail(self.max_files, F, self.G, self.step, self.latency, self.tau,
                                        num_threads=self.threads)
        self._F = F
        self._max_turns = self._tails.max_turns
        n = self.engine.max_inflight + 1
        self._stage = [torch.empty((self.rows, self.S), dtype=torch.float32, device=self.device) for _ in range(n)]
        self._turns = [np.empty((self.rows, self._max_turns, 3), dtype=np.float64) for _ in range(2)]
        self._nturns = [np.empty(self.rows, dtype=np.int32) for _ in range(2)]

    def __del__(self):
        try:
            for h in self._clu:
                self._lib.dz_clu_destroy(h)
        except Exception:
            pass

    # ------------------------------------------------------------------ the run
    class _PinnedPool:
        """Pinned host buffers for the files in flight, reused (allocating pinned memory costs
        milliseconds per file; the loader thread acquires, the run loop releases)."""

        def __init__(self):
            import threading
            self._free: List[torch.Tensor] = []
            self._lock = threading.Lock()

        def acquire(self, n: int) -> torch.Tensor:
            with self._lock:
                for i, t in enumerate(self._free):
                    if t.numel() >= n:
                        return self._free.pop(i)
            return torch.empty(max(1, int(n * 1.1)), dtype=torch.float32).pin_memory()

        def release(self, t: torch.Tensor) -> None:
            with self._lock:
                self._free.append(t)

    def host_buffer(self, n: int) -> np.ndarray:
        """A pinned float32 array of n samples from the pool, for a feeder that wants to decode a file
        straight into upload-ready memory (``Benchmark.run_batched``); hand it to ``run`` as the
        waveform and the loader skips its own copy."""
        t = self._pool.acquire(n)
        a = t.numpy()[:n]
        self._lent[a.__array_interface__["data"][0]] = t
        return a

    class _Loader:
        """Reads / pads the next files on a background thread while the GPU works (reading a WAV and
        converting it to float32 takes ~10 ms per 5 minutes of audio: done serially in front of the
        first launch it cost more than the GPU time of the whole file).  ``get(block)`` ->
        (uri, pinned float32 tensor, samples, shift) in the order of ``files``, None when nothing is
        ready yet (``block=False``), ``StopIteration`` at the end."""

        def __init__(self, files, owner: "FileBatch", depth: int = 4):
            import queue
            import threading
            self.q: "queue.Queue" = queue.Queue(maxsize=depth)
            self.done = False
            self.stop = False               # set by close(): the thread stops producing and hands its buffer back
            self._END = object()
            self._owner = owner

            def put(item) -> bool:          # q.put that gives up when the consumer has gone away
                while not self.stop:
                    try:
                        self.q.put(item, timeout=0.05)
                        return True
                    except queue.Full:
                        continue
                return False

            def work():
                try:
                    for uri, wav, shift in files:
                        lent = None
                        if isinstance(wav, np.ndarray) and wav.dtype == np.float32 and wav.ndim == 1:
                            lent = owner._lent.pop(wav.__array_interface__["data"][0], None)
                        if lent is not None:            # decoded straight into a pool buffer
                            t = lent
                        else:
                            wav = np.ascontiguousarray(wav, dtype=np.float32).reshape(-1)
                            t = owner._pool.acquire(len(wav))
                            t[:len(wav)].copy_(torch.from_numpy(wav))
                        if not put((uri, t, len(wav), shift)):
                            owner._pool.release(t)
                            return
                    put(self._END)
                except BaseException as exc:       # surfaces in the consumer
                    put(exc)

            threading.Thread(target=work, name="dz-file-loader", daemon=True).start()

        def close(self) -> None:
            """Stop the loader thread and give the pinned buffers of files that were read but never admitted
            back to the pool (the consumer is leaving early: an exception in a step)."""
            import queue
            self.stop = True
            while True:
                try:
                    item = self.q.get(timeout=0.1)
                except queue.Empty:
                    break
                if isinstance(item, tuple):
                    self._owner._pool.release(item[1])
            self.done = True

        def get(self, block: bool):
            import queue
            if self.done:
                raise StopIteration
            try:
                item = self.q.get(block=block)
            except queue.Empty:
                return None
            if item is self._END:
                self.done = True
                raise StopIteration
            if isinstance(item, BaseException):
                self.done = True
                raise item
            return item

    def run(self, files) -> dict:
        files = FileBatch._Loader(files, self)
        F = self.engine.seg.to(self.device).num_frames(self.S)
        self._ensure_state(F)
        K, D = None, self.engine.emb.dimension
        res = self.duration / F
        open_files: List[Optional[dict]] = [None] * self.max_files
        done: dict = {}
        exhausted = False
        inflight: List[tuple] = []
        step_no = 0

        def admit():
            """Fill free slots with files that are READY; wait for one only when nothing is open."""
            nonlocal exhausted
            for slot in range(self.max_files):
                if exhausted:
                    return
                if open_files[slot] is None:
                    try:
                        item = files.get(block=not any(f is not None for f in open_files))
                    except StopIteration:
                        exhausted = True
                        return
                    if item is None:
                        return
                    uri, pinned, nsamp, shift = item
                    if uri in done or any(f is not None and f["uri"] == uri for f in open_files):
                        self._pool.release(pinned)
                        raise ValueError(f"FileBatch.run: two files share the uri {uri!r}; the results are keyed by it")
                    nwin = (nsamp - self.S) // self.hop + 1 if nsamp >= self.S else 0
                    _lib.check(self._lib.dz_clu_reset(self._clu[slot]), "dz_clu_reset")
                    self._tails.reset(slot)
                    if nwin <= 0:
                        done[uri] = []
                        self._pool.release(pinned)      # shorter than one window: nothing to upload
                        return admit()
                    open_files[slot] = dict(uri=uri, shift=float(shift), nwin=nwin, sent=0, got=0, turns=[], start=0,
                                            audio=pinned[:nsamp].to(self.device, non_blocking=True), host=pinned)

        def launch_step():
            nonlocal step_no
            active = [(slot, f) for slot, f in enumerate(open_files) if f is not None and f["sent"] < f["nwin"]]
            if not active:
                return False
            per = max(1, self.rows // len(active))
            stage = self._stage[step_no % len(self._stage)]
            plan, r = [], 0
            for slot, f in active:
                c = min(per, f["nwin"] - f["sent"], self.rows - r)
                if c <= 0:
                    break
                w0 = f["sent"]
                # c consecutive windows of this file: a strided view of its resident audio -> dense rows
                view = f["audio"][w0 * self.hop: w0 * self.hop + (c - 1) * self.hop + self.S].unfold(0, self.S, self.hop)
                stage[r:r + c].copy_(view, non_blocking=True)
                # window start times by repeated addition, exactly as rearrange_audio_stream counts them
                # (`start += step`, operators.py:82-84): bit-identical for steps that are not dyadic
                starts = np.empty(c, dtype=np.float64)
                for j in range(c):
                    starts[j] = f["start"]
                    f["start"] += self.step
                plan.append((slot, f, r, c, starts))
                f["sent"] += c
                r += c
            ticket = self.engine._launch_rows(stage.data_ptr(), stage.stride(0), r, self.S, keep=stage)
            inflight.append((ticket, plan, r))
            step_no += 1
            return True

        def finish_step():
            ticket, plan, r = inflight.pop(0)
            self.engine._wait(ticket)
            try:
                _lib.range_check(self.device.index)
                seg = ticket["seg_h"].numpy()[:r]
                emb = ticket["emb_h"].numpy()[:r]
                k = len(plan)
                clus = (_lib.vp * k)(*[self._clu[slot] for slot, *_ in plan])
                tails = (_lib.vp * k)(*[self._tails._hs[slot] for slot, *_ in plan])
                row0 = np.array([p[2] for p in plan], dtype=np.int32)
                count = np.array([p[3] for p in plan], dtype=np.int32)
                starts = np.concatenate([p[4] for p in plan]).astype(np.float64)
                turns, nturns = self._turns[0], self._nturns[0]
                _lib.check(self._lib.dz_file_step_batch(
                    clus, tails, k, row0.ctypes.data, count.ctypes.data, seg.ctypes.data, seg.shape[1], seg.shape[2],
                    emb.ctypes.data, emb.shape[2], self.G, starts.ctypes.data, float(res), turns.ctypes.data,
                    self._max_turns, nturns.ctypes.data, None, self.threads), "dz_file_step_batch")
            finally:
                ticket["busy"] = False
                ticket["keep"] = None
            for slot, f, r0, c, _ in plan:
                for j in range(r0, r0 + c):
                    if nturns[j]:
                        f["turns"].append(turns[j, :nturns[j]].copy())
                f["got"] += c
                self.chunks_done += c
                if f["got"] == f["nwin"]:
                    done[f["uri"]] = (f["turns"], f["shift"])
                    open_files[slot] = None
                    self._pool.release(f["host"])     # its upload finished long ago (results depend on it)

        try:
            admit()
            while True:
                while len(inflight) < self.engine.max_inflight and launch_step():
                    pass
                if not inflight:
                    admit()
                    if not any(f is not None for f in open_files):
                        break
                    continue
                finish_step()
                admit()
        except BaseException:
            # leave nothing behind: tickets still on the GPU are waited for and handed back (a busy
            # ticket makes every later launch allocate pinned memory under running kernels), the open files'
            # and the loader's pinned buffers return to the pool, the loader thread stops
            for ticket, _, _ in inflight:
                try:
                    self.engine._wait(ticket)
                finally:
                    ticket["busy"], ticket["keep"] = False, None
            # the open files' pinned buffers may still be the source of a non_blocking upload: only a finished
            # copy makes them safe to hand to the next run's loader thread
            try:
                torch.cuda.synchronize(self.engine.device)
            except Exception:      # noqa: BLE001 — the original exception is the one to report
                pass
            for f in open_files:
                if f is not None:
                    self._pool.release(f["host"])
            files.close()
            raise
        out = {}
        for uri, rec in done.items():
            ann = Annotation(uri=uri, modality="speech")
            if rec:
                chunks, shift = rec
                for arr in chunks:
                    for s, e, g in arr:
                        ann[Segment(s + shift, e + shift), int(g)] = f"speaker{int(g)}"
            out[uri] = ann.support(self.patch_collar)
        return out
