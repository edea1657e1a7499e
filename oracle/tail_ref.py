"""Literal numpy restatement of the reference's per-chunk tail (TEST INFRASTRUCTURE).

* ``DelayedAggregationRef``  /root/reference/src/diart/blocks/aggregation.py:120-218 with the three
  strategies of :73-118 (crop every buffer and a Hamming window aligned to it, stack, weight)
* ``binarize_ref``           /root/reference/src/diart/blocks/utils.py:43-59 (frame loop)
* ``TailRef``                the buffer handling of /root/reference/src/diart/blocks/diarization.py:203-232

built on the ``pyannote.core`` stand-ins of ``oracle/pyannote_stub.py`` (the cropping rule is
pyannote.core's, restated there).  Pinned by ``tests/golden/tail.npz`` — outputs of the
reference's OWN aggregation.py / utils.py run on the same stand-ins.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from .pyannote_stub import Segment, SlidingWindow, SlidingWindowFeature


class DelayedAggregationRef:
    def __init__(self, step: float, latency: float = None, strategy: str = "hamming",
                 cropping_mode: str = "loose"):
        self.step, self.latency = step, step if latency is None else latency
        self.strategy, self.cropping_mode = strategy, cropping_mode
        self.num_overlapping_windows = int(round(self.latency / self.step))

    def _aggregate(self, buffers: List[SlidingWindowFeature], focus: Segment) -> np.ndarray:
        mode = self.cropping_mode
        if self.strategy == "first":
            return buffers[0].crop(focus, mode=mode, fixed=focus.duration)
        if self.strategy == "mean":
            return np.mean(np.stack([b.crop(focus, mode=mode, fixed=focus.duration) for b in buffers]), axis=0)
        num_frames = buffers[0].data.shape[0]
        hamming, intersection = [], []
        for buffer in buffers:
            b = buffer.crop(focus, mode=mode, fixed=focus.duration)
            h = SlidingWindowFeature(np.expand_dims(np.hamming(num_frames), axis=-1), buffer.sliding_window)
            hamming.append(h.crop(focus, mode=mode, fixed=focus.duration))
            intersection.append(b)
        hamming, intersection = np.stack(hamming), np.stack(intersection)
        return np.sum(hamming * intersection, axis=0) / np.sum(hamming, axis=0)

    def __call__(self, buffers: List[SlidingWindowFeature]) -> SlidingWindowFeature:
        start = buffers[-1].extent.end - self.latency
        region = Segment(start, start + self.step)
        data = self._aggregate(buffers, region)
        res = region.duration / data.shape[0]
        out = SlidingWindowFeature(data, SlidingWindow(start=region.start, duration=res, step=res))
        if len(buffers) == 1 and buffers[-1].extent.start == 0:       # aggregation.py:188-211
            num_frames = out.data.shape[0]
            first_region = Segment(0, region.end)
            first = buffers[0].crop(first_region, mode=self.cropping_mode, fixed=first_region.duration)
            first[-num_frames:] = out.data
            res = region.end / first.shape[0]
            out = SlidingWindowFeature(first, SlidingWindow(start=0, duration=res, step=res))
        return out


def binarize_ref(segmentation: SlidingWindowFeature, threshold: float) -> List[Tuple[float, float, int]]:
    """-> [(start, end, speaker)] sorted by (start, end, speaker)."""
    num_frames, num_speakers = segmentation.data.shape
    timestamps = segmentation.sliding_window
    is_active = segmentation.data > threshold
    is_active = np.append(is_active, [[False] * num_speakers], axis=0)
    start_times = np.zeros(num_speakers) + timestamps[0].middle
    turns = []
    for t in range(num_frames):
        onsets = np.logical_and(np.logical_not(is_active[t]), is_active[t + 1])
        start_times[onsets] = timestamps[t + 1].middle
        offsets = np.logical_and(is_active[t], np.logical_not(is_active[t + 1]))
        for spk in np.where(offsets)[0]:
            s, e = float(start_times[spk]), float(timestamps[t + 1].middle)
            if e - s > 1e-6:
                turns.append((s, e, int(spk)))
    return sorted(turns)


class TailRef:
    """scores (F, G) of consecutive chunks of one stream -> turns of the region that became final."""

    def __init__(self, step: float, latency: float, tau_active: float):
        self.agg = DelayedAggregationRef(step, latency, "hamming", "loose")
        self.tau = tau_active
        self.buffer: List[SlidingWindowFeature] = []

    def __call__(self, scores: SlidingWindowFeature):
        self.buffer.append(scores)
        agg = self.agg(self.buffer)
        turns = binarize_ref(agg, self.tau)
        if len(self.buffer) == self.agg.num_overlapping_windows:
            self.buffer = self.buffer[1:]
        return agg, turns
