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


def _crop(data: np.ndarray, buffer: SlidingWindowFeature, focus: Segment, mode: str) -> np.ndarray:
    """``data[_rows(buffer, focus, mode)]`` (a fresh array, like ``crop``): a contiguous range that lies inside the
    buffer — every call but the first / last of a stream — is one memcpy instead of an 8000-element gather (the
    audio passthrough of a 500 ms region was 50 us of a pipeline call's ~75 us host time per chunk)."""
    first, last = buffer.sliding_window.crop_range(focus, mode, fixed=focus.duration)
    if 0 <= first <= last <= data.shape[0]:
        return data[first:last].copy()
    return data[np.clip(np.arange(first, last), 0, data.shape[0] - 1)]


class AggregationStrategy:
    """hamming | mean | first over the frames of ``focus`` shared by the buffers."""

    def __init__(self, name: str = "hamming", cropping_mode: str = "loose"):
        assert name in ("mean", "hamming", "first")
        assert cropping_mode in _MODES, f"Invalid cropping mode `{cropping_mode}`"
        self.name, self.cropping_mode = name, cropping_mode
        self._hamming = {}

    @staticmethod
    def build(name: str, cropping_mode: str = "loose") -> "AggregationStrategy":
        """The strategy class of that name, like the reference's factory (aggregation.py:28-40)."""
        assert name in ("mean", "hamming", "first")
        return {"mean": AverageStrategy, "hamming": HammingWeightedAverageStrategy,
                "first": FirstOnlyStrategy}[name](cropping_mode)

    def aggregate(self, buffers: List[SlidingWindowFeature], focus: Segment) -> np.ndarray:
        if self.name == "first":
            return _crop(buffers[0].data, buffers[0], focus, self.cropping_mode)
        crops = [_crop(b.data, b, focus, self.cropping_mode) for b in buffers]
        if self.name == "mean":
            return np.mean(np.stack(crops), axis=0)
        num_frames = buffers[0].data.shape[0]
        h = self._hamming.get(num_frames)
        if h is None:
            h = self._hamming[num_frames] = np.expand_dims(np.hamming(num_frames), axis=-1)
        hamming = np.stack([_crop(h, b, focus, self.cropping_mode) for b in buffers])
        return np.sum(hamming * np.stack(crops), axis=0) / np.sum(hamming, axis=0)

    def __call__(self, buffers: List[SlidingWindowFeature], focus: Segment) -> SlidingWindowFeature:
        aggregation = self.aggregate(buffers, focus)
        res = focus.duration / aggregation.shape[0]
        return SlidingWindowFeature(aggregation, SlidingWindow(start=focus.start, duration=res, step=res))


class HammingWeightedAverageStrategy(AggregationStrategy):
    """Average weighted by the Hamming window aligned to each buffer (aggregation.py:95-118)."""

    def __init__(self, cropping_mode: str = "loose"):
        super().__init__("hamming", cropping_mode)


class AverageStrategy(AggregationStrategy):
    """Simple average over the focus region (aggregation.py:73-92)."""

    def __init__(self, cropping_mode: str = "loose"):
        super().__init__("mean", cropping_mode)


class FirstOnlyStrategy(AggregationStrategy):
    """Keep the first buffer that covers the region (aggregation.py:60-70)."""

    def __init__(self, cropping_mode: str = "loose"):
        super().__init__("first", cropping_mode)


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


_STRATEGY = {"hamming": 0, "mean": 1, "first": 2}
_MODE = {"strict": 0, "loose": 1, "center": 2}


class BatchedOutputTail:
    """``DelayedAggregation`` + ``Binarize`` of N independent streams in C++ (``dz_tail_step_batch``,
    host threads, fp64): what ``SpeakerDiarization.__call__`` does after clustering for one stream
    (reference ``blocks/diarization.py:203-232``), for all the streams of a ``StreamBatch`` at once
    and bit-identical to the Python blocks above (``tests/test_tail.py``).

    ``__call__(scores (N,F,G) f64, chunk_start (N,), resolution (N,) | float)`` ->
    ``(aggregated, rows, t0, res, turns, nturns)``: ``aggregated[i, :rows[i]]`` are the scores of the
    region ``[t0[i], t0[i] + rows[i] * res[i])``; ``turns[i, :nturns[i]]`` = (start, end, speaker)."""

    def __init__(self, num_streams: int, frames: int, speakers: int, step: float,
                 latency: Optional[float] = None, threshold: float = 0.5, strategy: str = "hamming",
                 cropping_mode: str = "loose", num_threads: int = 8, max_turns: Optional[int] = None):
        from .. import _lib
        import ctypes as C
        self._lib, self._C = _lib, C
        latency = step if latency is None else latency
        assert step <= latency, "Invalid latency requested"
        assert strategy in _STRATEGY and cropping_mode in _MODE
        self.n, self.F, self.G = num_streams, frames, speakers
        self.num_threads = num_threads
        self.num_overlapping_windows = int(round(latency / step))
        hamming = np.ascontiguousarray(np.hamming(frames), dtype=np.float64)
        lib = _lib.load()
        hs = []
        for _ in range(num_streams):
            h = _lib.vp()
            _lib.check(lib.dz_tail_create(frames, speakers, float(step), float(latency), float(threshold),
                                          _STRATEGY[strategy], _MODE[cropping_mode],
                                          hamming.ctypes.data, C.byref(h)), "dz_tail_create")
            hs.append(h)
        self._hs = hs
        self._handles = (_lib.vp * num_streams)(*hs)
        self.max_rows = frames + 2
        self.max_turns = max_turns if max_turns is not None else speakers * (self.max_rows // 2 + 1)
        n = num_streams
        self._agg = np.empty((n, self.max_rows, speakers), dtype=np.float64)
        self._rows = np.empty(n, dtype=np.int32)
        self._t0 = np.empty(n, dtype=np.float64)
        self._res = np.empty(n, dtype=np.float64)
        self._turns = np.empty((n, self.max_turns, 3), dtype=np.float64)
        self._nturns = np.empty(n, dtype=np.int32)

    def reset(self, slot=None):
        for h in (self._hs if slot is None else [self._hs[slot]]):
            self._lib.check(self._lib.load().dz_tail_reset(h), "dz_tail_reset")

    def __del__(self):
        try:
            lib = self._lib.load()
            for h in self._hs:
                lib.dz_tail_destroy(h)
        except Exception:
            pass

    def __call__(self, scores: np.ndarray, chunk_start, resolution, slots=None):
        """``slots``: the streams the rows of ``scores`` belong to (default: all, in order); the
        outputs are then indexed by row, not by slot."""
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        n = self.n if slots is None else len(slots)
        handles = self._handles if slots is None else (self._lib.vp * n)(*[self._hs[i] for i in slots])
        assert scores.shape == (n, self.F, self.G), scores.shape
        start = np.ascontiguousarray(np.broadcast_to(np.asarray(chunk_start, dtype=np.float64), (n,)))
        res = np.ascontiguousarray(np.broadcast_to(np.asarray(resolution, dtype=np.float64), (n,)))
        self._lib.check(self._lib.load().dz_tail_step_batch(
            handles, n, scores.ctypes.data, start.ctypes.data, res.ctypes.data,
            self._agg.ctypes.data, self._rows.ctypes.data, self._t0.ctypes.data, self._res.ctypes.data,
            self._turns.ctypes.data, self.max_turns, self._nturns.ctypes.data, self.num_threads),
            "dz_tail_step_batch")
        return self._agg, self._rows, self._t0, self._res, self._turns, self._nturns

    @staticmethod
    def annotation(turns: np.ndarray, nturns: int, uri=None, shift: float = 0.0):
        """Speech turns of one stream as the ``Annotation`` ``Binarize`` builds (utils.py:47-58)."""
        from ..features import Annotation
        ann = Annotation(uri=uri, modality="speech")
        for s, e, g in turns[:nturns]:
            ann[Segment(s + shift, e + shift), int(g)] = f"speaker{int(g)}"
        return ann
