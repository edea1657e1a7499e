"""``OnlineSpeakerClustering``: constrained incremental clustering of speaker embeddings.

Same constructor, call signature and public state (``centers``, ``active_centers``,
``blocked_centers``) as ``/root/reference/src/diart/blocks/clustering.py:10-218``.  The
decision logic (cosine distances in fp64, rectangular Hungarian with the 1e10 cannot-link
sentinel, delta_new thresholding, centroid creation / update) runs in ``libdiart_amd.so``
(``csrc/cluster.cpp``); this class only marshals arrays.  ``BatchedSpeakerClustering`` drives
N independent streams at once (the reference has one pipeline per stream,
``/root/reference/src/diart/blocks/diarization.py:146-155``).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Set

import numpy as np
import torch

from .. import _lib
from ..features import SlidingWindowFeature


class SpeakerAssignment:
    """Result of ``identify``: which global speaker each local speaker was mapped to
    (the information a reference ``SpeakerMap`` carries at this point)."""

    def __init__(self, assignment: np.ndarray, num_global: int):
        self.assignment = assignment        # (K,) global index or -1
        self.num_global = num_global

    def valid_assignments(self):
        src = [int(s) for s in np.nonzero(self.assignment >= 0)[0]]
        return src, [int(self.assignment[s]) for s in src]

    def apply(self, source_scores: np.ndarray) -> np.ndarray:
        out = np.zeros((source_scores.shape[0], self.num_global))
        for s, t in zip(*self.valid_assignments()):
            out[:, t] = source_scores[:, s]
        return out


class OnlineSpeakerClustering:
    def __init__(self, tau_active: float, rho_update: float, delta_new: float,
                 metric: Optional[str] = "cosine", max_speakers: int = 20):
        if metric != "cosine":
            raise ValueError("only the cosine metric of the diarization pipeline is built "
                             "(reference diarization.py:149-153)")
        self.tau_active, self.rho_update, self.delta_new = tau_active, rho_update, delta_new
        self.metric, self.max_speakers = metric, max_speakers
        self._h = _lib.vp()
        _lib.check(_lib.load().dz_clu_create(float(tau_active), float(rho_update), float(delta_new),
                                             int(max_speakers), C.byref(self._h)), "dz_clu_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().dz_clu_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- state, as the reference exposes it ------------------------------------------
    @property
    def centers(self) -> Optional[np.ndarray]:
        dim = _lib.load().dz_clu_dim(self._h)
        if dim == 0:
            return None
        out = np.empty((self.max_speakers, dim), dtype=np.float64)
        _lib.check(_lib.load().dz_clu_get_centers(self._h, out.ctypes.data, dim), "dz_clu_get_centers")
        return out

    @property
    def active_centers(self) -> Set[int]:
        mask = np.zeros(self.max_speakers, dtype=np.int32)
        _lib.check(_lib.load().dz_clu_get_active(self._h, mask.ctypes.data), "dz_clu_get_active")
        return set(int(i) for i in np.nonzero(mask)[0])

    @property
    def blocked_centers(self) -> Set[int]:
        return set()  # never populated by the reference either (clustering.py:46,83)

    @property
    def num_known_speakers(self) -> int:
        return len(self.active_centers)

    @property
    def num_blocked_speakers(self) -> int:
        return 0

    @property
    def num_free_centers(self) -> int:
        return self.max_speakers - self.num_known_speakers

    @property
    def inactive_centers(self) -> List[int]:
        act = self.active_centers
        return [c for c in range(self.max_speakers) if c not in act]

    def reset(self):
        _lib.check(_lib.load().dz_clu_reset(self._h), "dz_clu_reset")

    # ---- one chunk ----------------------------------------------------------------------
    def _step(self, seg: np.ndarray, emb: np.ndarray, want_scores: bool):
        seg = np.ascontiguousarray(seg, dtype=np.float32)
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        if seg.ndim != 2 or emb.ndim != 2 or emb.shape[0] != seg.shape[1]:
            raise ValueError(f"expected segmentation (frames, speakers) and embeddings "
                             f"(speakers, dim), got {seg.shape} and {emb.shape}")
        F, K = seg.shape
        scores = np.empty((F, self.max_speakers), dtype=np.float64) if want_scores else None
        assign = np.empty(K, dtype=np.int32)
        rc = _lib.load().dz_clu_step(self._h, seg.ctypes.data, F, K, emb.ctypes.data, emb.shape[1],
                                     scores.ctypes.data if want_scores else None, assign.ctypes.data)
        if rc == 3:  # the reference raises here too (assert / scipy ValueError)
            raise AssertionError(_lib.load().dz_last_error().decode())
        _lib.check(rc, "dz_clu_step")
        return scores, assign

    def identify(self, segmentation: SlidingWindowFeature, embeddings: torch.Tensor) -> SpeakerAssignment:
        emb = embeddings.detach().cpu().numpy() if isinstance(embeddings, torch.Tensor) else embeddings
        _, assign = self._step(segmentation.data, emb, False)
        return SpeakerAssignment(assign.astype(np.int64), self.max_speakers)

    def __call__(self, segmentation: SlidingWindowFeature, embeddings: torch.Tensor) -> SlidingWindowFeature:
        emb = embeddings.detach().cpu().numpy() if isinstance(embeddings, torch.Tensor) else embeddings
        scores, _ = self._step(segmentation.data, emb, True)
        return SlidingWindowFeature(scores, segmentation.sliding_window)


# the north-star's name for the same block
IncrementalSpeakerClustering = OnlineSpeakerClustering


class BatchedSpeakerClustering:
    """N independent clustering states stepped together on host threads
    (``dz_clu_step_batch``): seg (N,F,K) f32, emb (N,K,D) f32 -> scores (N,F,G) f64, assign (N,K)."""

    def __init__(self, num_streams: int, tau_active: float, rho_update: float, delta_new: float,
                 max_speakers: int = 20, num_threads: int = 8):
        self.streams = [OnlineSpeakerClustering(tau_active, rho_update, delta_new, "cosine", max_speakers)
                        for _ in range(num_streams)]
        self.max_speakers, self.num_threads = max_speakers, num_threads
        self._handles = (_lib.vp * num_streams)(*[s._h for s in self.streams])

    def reset(self, slot=None):
        for s in (self.streams if slot is None else [self.streams[slot]]):
            s.reset()

    def __call__(self, seg: np.ndarray, emb: np.ndarray, want_scores: bool = True, slots=None):
        """``slots``: the streams the N rows belong to (default: all, in order)."""
        seg = np.ascontiguousarray(seg, dtype=np.float32)
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        N, F, K = seg.shape
        handles = self._handles
        if slots is not None:
            handles = (_lib.vp * N)(*[self.streams[i]._h for i in slots])
        assert N == len(handles) and emb.shape[:2] == (N, K)
        scores = np.empty((N, F, self.max_speakers), dtype=np.float64) if want_scores else None
        assign = np.empty((N, K), dtype=np.int32)
        rc = _lib.load().dz_clu_step_batch(handles, N, seg.ctypes.data, F, K, emb.ctypes.data,
                                           emb.shape[2], scores.ctypes.data if want_scores else None,
                                           assign.ctypes.data, self.num_threads)
        if rc == 3:
            raise AssertionError(_lib.load().dz_last_error().decode())
        _lib.check(rc, "dz_clu_step_batch")
        return scores, assign
