"""Cross-stream driver of the hot path: N concurrent audio streams, one chunk each per step.

The reference batches consecutive windows of ONE stream (``inference.py:126-128``) and keeps one
pipeline object per stream (``blocks/diarization.py:121-125``); 64 concurrent real-time streams
(BASELINE.json config 2) therefore need a driver the reference does not have.  ``StreamBatch``
stacks the current window of every stream into one segmentation / embedding call and steps N
independent clustering states, producing for each stream exactly what its own
``SpeakerDiarization.__call__`` computes at lines 186-203: segmentation, overlap-aware
normalised embeddings and the permuted ``(frames, max_speakers)`` scores.

GPU schedule per step (two HIP streams):

    stream A : dz_seg_forward (SincNet -> 4 x {x-projection GEMM, persistent LSTM} -> MLP)
               -> dz_osp  --event-->
    stream B : dz_emb_frames (SincNet -> 5 TDNN; independent of the segmentation, fills the
               CUs the latency-bound LSTM leaves idle)  <--wait--  dz_emb_pool -> D2H (pinned)
    host     : clustering of step t-1 (C++ threads, fp64) while the GPU runs step t
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .blocks.clustering import BatchedSpeakerClustering
from .models import HipEmbedding, HipSegmentation, _as_rows


class StreamBatch:
    def __init__(self, segmentation: HipSegmentation, embedding: HipEmbedding, num_streams: int,
                 tau_active: float = 0.6, rho_update: float = 0.3, delta_new: float = 1.0,
                 gamma: float = 3, beta: float = 10, max_speakers: int = 20,
                 normalize_embedding_weights: bool = False,
                 device: Optional[torch.device] = None, cluster_threads: int = 8):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.seg, self.emb = segmentation.to(self.device), embedding.to(self.device)
        self.device = self.seg.device
        self.n = num_streams
        self.gamma, self.beta, self.norm_w = float(gamma), float(beta), bool(normalize_embedding_weights)
        self.clustering = BatchedSpeakerClustering(num_streams, tau_active, rho_update, delta_new,
                                                   max_speakers, cluster_threads)
        self.max_speakers = max_speakers
        self.stream_a = torch.cuda.Stream(self.device)
        self.stream_b = torch.cuda.Stream(self.device)
        self._slots: List[dict] = []
        self._lib = _lib.load()
        self._ctx = _lib.context(self.device.index)

    def reset(self):
        self.clustering.reset()

    # ------------------------------------------------------------------ GPU half
    def _slot(self, F: int, K: int, D: int) -> dict:
        for s in self._slots:
            if not s["busy"] and s["shape"] == (F, K, D):
                return s
        n, dev = self.n, self.device
        s = dict(shape=(F, K, D), busy=False,
                 seg=torch.empty((n, F, K), dtype=torch.float32, device=dev),
                 w=torch.empty((n, K, F), dtype=torch.float32, device=dev),
                 emb=torch.empty((n, K, D), dtype=torch.float32, device=dev),
                 seg_h=torch.empty((n, F, K), dtype=torch.float32).pin_memory(),
                 emb_h=torch.empty((n, K, D), dtype=torch.float32).pin_memory(),
                 ev_seg=torch.cuda.Event(), ev_in=torch.cuda.Event(), done=torch.cuda.Event())
        self._slots.append(s)
        return s

    def launch(self, waves: torch.Tensor) -> dict:
        """Enqueue the GPU work for one step.  ``waves``: (N, S) or (N, 1, S) float32 on the GPU;
        a strided rolling-window view is used in place.  Returns a ticket for ``finish``."""
        rows = _as_rows(waves)
        N, S = rows.shape
        assert N == self.n, f"expected {self.n} streams, got {N}"
        F, K, D = self.seg.num_frames(S), None, self.emb.dimension
        hseg, hemb = self.seg._need(S, N), self.emb._need(S, N)
        K = self.seg.num_speakers
        slot = self._slot(F, K, D)
        slot["busy"] = True
        lib, stride = self._lib, (rows.stride(0) if N > 1 else S)
        cur = torch.cuda.current_stream(self.device)
        slot["ev_in"].record(cur)                       # inputs produced on the caller's stream
        a, b = self.stream_a, self.stream_b
        a.wait_event(slot["ev_in"])
        b.wait_event(slot["ev_in"])
        _lib.check(lib.dz_seg_forward(hseg, rows.data_ptr(), stride, N, slot["seg"].data_ptr(),
                                      a.cuda_stream), "dz_seg_forward")
        _lib.check(lib.dz_osp(self._ctx, slot["seg"].data_ptr(), N, F, K, self.gamma, self.beta,
                              int(self.norm_w), 1, slot["w"].data_ptr(), a.cuda_stream), "dz_osp")
        slot["ev_seg"].record(a)
        _lib.check(lib.dz_emb_frames(hemb, rows.data_ptr(), stride, N, b.cuda_stream), "dz_emb_frames")
        b.wait_event(slot["ev_seg"])
        _lib.check(lib.dz_emb_pool(hemb, slot["w"].data_ptr(), N, K, F, 1, slot["emb"].data_ptr(),
                                   b.cuda_stream), "dz_emb_pool")
        with torch.cuda.stream(b):
            slot["seg_h"].copy_(slot["seg"], non_blocking=True)
            slot["emb_h"].copy_(slot["emb"], non_blocking=True)
        slot["done"].record(b)
        slot["keep"] = rows                              # keep the view alive until the GPU is done
        return slot

    # ------------------------------------------------------------------ host half
    def finish(self, ticket: dict, want_scores: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """Wait for the step's GPU work, run the N clustering updates.
        -> (segmentation (N,F,K) f32, embeddings (N,K,D) f32, scores (N,F,G) f64 | None, assign (N,K))."""
        ticket["done"].synchronize()
        seg = ticket["seg_h"].numpy()
        emb = ticket["emb_h"].numpy()
        scores, assign = self.clustering(seg, emb, want_scores)
        ticket["busy"] = False
        ticket["keep"] = None
        return seg, emb, scores, assign

    def __call__(self, waves: torch.Tensor):
        return self.finish(self.launch(waves))
