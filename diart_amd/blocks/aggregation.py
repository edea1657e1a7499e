"""``DelayedAggregation``: combine the overlapping windows that cover the region
``[t - latency, t - latency + step)`` of a stream (reference:
``/root/reference/src/diart/blocks/aggregation.py`` — strategies :73-118, ``DelayedAggregation``
:120-218, first-chunk prepend :188-211).

Same constructor, call signature and numbers as the reference.  Instead of wrapping every
buffer and a fresh Hamming window in ``SlidingWindowFeature`` objects and cropping each, the
frame range of the focus region is computed once per buffer from the frame grid (the
``pyannote.core`` crop rule, ``diart_amd.features.SlidingWindow.crop_range``) and rows are
gathered with clipped indices (rows requested outside a buffer repeat its first / last row,
exactly what ``crop(..., fixed=...)`` does).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from ..features import Segment, SlidingWindow, SlidingWindowFeature

_MODES = ("strict", "loose", "center")


def _rows(buffer: SlidingWindowFeature, focus: Segment, mode: str) -> np.ndarray:
    first, last = buffer.sliding_window.crop_range(focus, mode, fixed=focus.duration)
    return np.clip(np.arange(first, last), 0, buffer.data.shape[0] - 1)


class AggregationStrategy:
    """hamming | mean | first over the frames of ``focus`` shared by the buffers."""

    def __init__(self, name: str = "hamming", cropping_mode: str = "loose"):
        assert name in ("mean", "hamming", "first")
        assert cropping_mode in _MODES, f"Invalid cropping mode `{cropping_mode}`"
        self.name, self.cropping_mode = name, cropping_mode
        self._hamming = {}

    @staticmethod
    def build(name: str, cropping_mode: str = "loose") -> "AggregationStrategy":
        return AggregationStrategy(name, cropping_mode)

    def aggregate(self, buffers: List[SlidingWindowFeature], focus: Segment) -> np.ndarray:
        if self.name == "first":
            return buffers[0].data[_rows(buffers[0], focus, self.cropping_mode)]
        crops = [b.data[_rows(b, focus, self.cropping_mode)] for b in buffers]
        if self.name == "mean":
            return np.mean(np.stack(crops), axis=0)
        num_frames = buffers[0].data.shape[0]
        h = self._hamming.get(num_frames)
        if h is None:
            h = self._hamming[num_frames] = np.expand_dims(np.hamming(num_frames), axis=-1)
        hamming = np.stack([h[_rows(b, focus, self.cropping_mode)] for b in buffers])
        return np.sum(hamming * np.stack(crops), axis=0) / np.sum(hamming, axis=0)

    def __call__(self, buffers: List[SlidingWindowFeature], focus: Segment) -> SlidingWindowFeature:
        aggregation = self.aggregate(buffers, focus)
        res = focus.duration / aggregation.shape[0]
        return SlidingWindowFeature(aggregation, SlidingWindow(start=focus.start, duration=res, step=res))


class DelayedAggregation:
    def __init__(self, step: float, latency: Optional[float] = None, strategy: str = "hamming",
                 cropping_mode: str = "loose"):
        self.step, self.latency, self.strategy = step, latency, strategy
        assert cropping_mode in _MODES, f"Invalid cropping mode `{cropping_mode}`"
        self.cropping_mode = cropping_mode
        if self.latency is None:
            self.latency = self.step
        assert self.step <= self.latency, "Invalid latency requested"
        self.num_overlapping_windows = int(round(self.latency / self.step))
        self.aggregate = AggregationStrategy.build(self.strategy, self.cropping_mode)

    def _prepend(self, output_window: SlidingWindowFeature, output_region: Segment,
                 buffers: List[SlidingWindowFeature]) -> SlidingWindowFeature:
        # first buffer of a stream: output everything up to the end of the region (aggregation.py:188-211)
        if len(buffers) == 1 and buffers[-1].extent.start == 0:
            num_frames = output_window.data.shape[0]
            first_region = Segment(0, output_region.end)
            first_output = buffers[0].crop(first_region, mode=self.cropping_mode,
                                           fixed=first_region.duration).copy()
            first_output[-num_frames:] = output_window.data
            res = output_region.end / first_output.shape[0]
            output_window = SlidingWindowFeature(first_output, SlidingWindow(start=0, duration=res, step=res))
        return output_window

    def __call__(self, buffers: List[SlidingWindowFeature]) -> SlidingWindowFeature:
        start = buffers[-1].extent.end - self.latency
        region = Segment(start, start + self.step)
        return self._prepend(self.aggregate(buffers, region), region, buffers)
