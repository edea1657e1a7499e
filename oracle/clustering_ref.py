"""numpy restatement of the reference's incremental speaker clustering (TEST INFRASTRUCTURE).

Follows ``/root/reference/src/diart/blocks/clustering.py`` (``identify`` :119-210, ``update``
:85-99, ``add_center`` :101-117, ``__call__`` :212-218) and the ``SpeakerMap`` operations it uses
from ``/root/reference/src/diart/mapping.py`` (``mapped_indices`` :18-21, ``hard_speaker_map``
:23-46, ``valid_assignments`` :217-231, ``set_source_speaker`` :245-251, ``unmap_threshold``
:260-273, ``unmap_speakers`` :275-294, ``apply`` :341-360), flattened into plain functions on a
``(K, G)`` cost matrix.  Third-party pieces are used as the reference uses them:
``scipy.optimize.linear_sum_assignment`` (mapping.py:8,16) and ``scipy.spatial.distance.cdist``
(= ``pyannote.core.utils.distance.cdist``, mapping.py:7,175).

Pinned by ``tests/golden/clustering_*.npz`` (outputs of the reference's own classes).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist

INVALID = 1e10  # MinimizationObjective.invalid_value, mapping.py:48-52


def _mapped_rows(matrix: np.ndarray) -> List[int]:
    return list(np.where(np.min(matrix, axis=1) != INVALID)[0])


def _valid_assignments(matrix: np.ndarray) -> Tuple[List[int], List[int]]:
    """mapping.py:217-231 (loose): enumerate(lsap columns), keep rows that are mapped."""
    raw = list(linear_sum_assignment(matrix, False)[1])
    mapped = _mapped_rows(matrix)
    src, tgt = [], []
    for s, t in enumerate(raw):
        if s in mapped:
            src.append(s)
            tgt.append(t)
    return src, tgt


class OnlineSpeakerClusteringRef:
    def __init__(self, tau_active: float, rho_update: float, delta_new: float,
                 metric: str = "cosine", max_speakers: int = 20):
        self.tau_active, self.rho_update, self.delta_new = tau_active, rho_update, delta_new
        self.metric, self.max_speakers = metric, max_speakers
        self.centers: Optional[np.ndarray] = None
        self.active_centers: set = set()
        self.blocked_centers: set = set()

    # clustering.py:68-71
    def _next_center(self) -> Optional[int]:
        for c in range(self.max_speakers):
            if c not in self.active_centers and c not in self.blocked_centers:
                return c
        return None

    def _add_center(self, embedding: np.ndarray) -> int:
        c = self._next_center()
        self.centers[c] = embedding
        self.active_centers.add(c)
        return c

    def identify(self, seg: np.ndarray, embeddings: np.ndarray) -> np.ndarray:
        """seg (F,K) float32, embeddings (K,D) float32 -> final (K,G) mapping matrix."""
        K, G = seg.shape[1], self.max_speakers
        active = np.where(np.max(seg, axis=0) >= self.tau_active)[0]
        long_spk = np.where(np.mean(seg, axis=0) >= self.rho_update)[0]
        no_nan = np.where(~np.isnan(embeddings).any(axis=1))[0]
        active = np.intersect1d(active, no_nan)

        if self.centers is None:  # :149-158
            self.centers = np.zeros((G, embeddings.shape[1]))
            self.active_centers, self.blocked_centers = set(), set()
            m = np.ones((K, G)) * INVALID
            for spk in active:
                m[spk, self._add_center(embeddings[spk])] = 0
            return m

        dist = cdist(embeddings, self.centers, metric=self.metric)  # :161
        inactive_centers = [c for c in range(G)
                            if c not in self.active_centers or c in self.blocked_centers]
        for spk in range(K):                                         # :163-166
            if spk not in active:
                dist[spk] = INVALID
        for c in inactive_centers:
            dist[:, c] = INVALID

        valid = dist.copy()                                          # :168 unmap_threshold
        for s, t in zip(*_valid_assignments(dist)):
            if dist[s, t] >= self.delta_new:
                valid[s] = INVALID

        mapped = _mapped_rows(valid)
        missed = [s for s in active if s not in mapped]              # :171-173
        new_center_speakers: List[int] = []
        num_free = G - len(self.active_centers) - len(self.blocked_centers)
        for spk in missed:                                           # :176-194
            has_space = len(new_center_speakers) < num_free
            if has_space and spk in long_spk:
                new_center_speakers.append(spk)
            else:
                pref = [g for g in np.argsort(dist[spk, :]) if g in self.active_centers]
                _, g_assigned = _valid_assignments(valid)
                free = [g for g in pref if g not in g_assigned]
                if free:
                    valid = valid.copy()
                    valid[spk, free[0]] = 0

        for ls, gs in zip(*_valid_assignments(valid)):               # :197-202
            if ls not in missed and ls in long_spk:
                assert gs in self.active_centers, "Cannot update unknown centers"
                self.centers[gs] += embeddings[ls]
        for spk in new_center_speakers:                              # :205-208
            valid = valid.copy()
            valid[spk, self._add_center(embeddings[spk])] = 0
        return valid

    def __call__(self, seg: np.ndarray, embeddings: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """-> (scores (F,G) float64, assignment (K,) int: global speaker or -1)."""
        m = self.identify(seg, np.asarray(embeddings))
        out = np.zeros((seg.shape[0], self.max_speakers))
        assign = -np.ones(seg.shape[1], dtype=np.int64)
        for s, t in zip(*_valid_assignments(m)):                     # mapping.py:341-360
            out[:, t] = seg[:, s]
            assign[s] = t
        return out, assign
