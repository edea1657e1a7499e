"""Temporal feature containers and the type-preserving formatter of the hot path.

``pyannote.core`` is used when installed; otherwise the two containers the path needs
(``SlidingWindow``, ``SlidingWindowFeature``) are provided here with the
same attributes, so blocks accept and return the same kinds of object as the reference's
(``/root/reference/src/diart/features.py:8``: ``SlidingWindowFeature | ndarray | Tensor``).
"""
from __future__ import annotations

from typing import Optional, Union

import numpy as np
import torch

try:  # pragma: no cover - depends on the environment
    from pyannote.core import Segment, SlidingWindow, SlidingWindowFeature
except ImportError:

    class Segment:
        def __init__(self, start: float, end: float):
            self.start, self.end = start, end

        @property
        def duration(self) -> float:
            return self.end - self.start

        @property
        def middle(self) -> float:
            return 0.5 * (self.start + self.end)

        def __iter__(self):
            yield self.start
            yield self.end

        def __repr__(self):
            return f"<Segment({self.start:g}, {self.end:g})>"

    class SlidingWindow:
        def __init__(self, duration: float = 0.030, step: float = 0.010, start: float = 0.0, end=None):
            self.duration, self.step, self.start, self.end = duration, step, start, end

        def __getitem__(self, i: int) -> Segment:
            s = self.start + i * self.step
            return Segment(s, s + self.duration)

    class SlidingWindowFeature:
        def __init__(self, data: np.ndarray, sliding_window: SlidingWindow):
            self.data, self.sliding_window = data, sliding_window

        def __len__(self):
            return self.data.shape[0]

        def __getitem__(self, i):
            return self.data[i]

        @property
        def extent(self) -> Segment:
            sw, n = self.sliding_window, self.data.shape[0]
            return Segment(sw.start, sw.start + (n - 1) * sw.step + sw.duration)


TemporalFeatures = Union[SlidingWindowFeature, np.ndarray, torch.Tensor]


class TemporalFeatureFormatter:
    """``cast`` -> float32 tensor ``(batch, frames, dim)``; ``restore_type`` gives features of
    the kind last cast (reference features.py:77-138).  A ``SlidingWindowFeature`` comes back as
    one with the resolution ``duration / frames`` and the start time of the input."""

    def __init__(self):
        self._kind: Optional[str] = None
        self._duration = 0.0
        self._start = 0.0

    def cast(self, features: TemporalFeatures) -> torch.Tensor:
        if isinstance(features, SlidingWindowFeature):
            sw = features.sliding_window
            assert sw.duration == sw.step, "Features sliding window duration and step must be equal"
            self._kind = "swf"
            self._duration = features.data.shape[0] * sw.duration
            self._start = sw.start
            data = torch.from_numpy(features.data)
        elif isinstance(features, np.ndarray):
            self._kind, data = "numpy", torch.from_numpy(features)
        elif isinstance(features, torch.Tensor):
            self._kind, data = "torch", features
        else:
            raise ValueError(
                "Unknown format. Provide one of SlidingWindowFeature, numpy.ndarray, torch.Tensor")
        assert data.ndim in (2, 3), "Temporal features must be 2D or 3D"
        if data.ndim == 2:
            data = data.unsqueeze(0)
        return data.float()

    def restore_type(self, features: torch.Tensor) -> TemporalFeatures:
        if self._kind == "torch":
            return features
        if self._kind == "numpy":
            return features.cpu().numpy()
        if self._kind == "swf":
            batch, frames, _ = features.shape
            assert batch == 1, "Batched SlidingWindowFeature objects are not supported"
            res = self._duration / frames
            return SlidingWindowFeature(features.squeeze(dim=0).cpu().numpy(),
                                        SlidingWindow(start=self._start, duration=res, step=res))
        raise RuntimeError("restore_type() called before cast()")
