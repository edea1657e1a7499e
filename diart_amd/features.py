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
    from pyannote.core import Annotation, Segment, SlidingWindow, SlidingWindowFeature, Timeline
except ImportError:

    class Segment:
        """Time interval [start, end) in seconds (pyannote.core.Segment surface used by diart)."""

        __slots__ = ("start", "end")

        def __init__(self, start: float = 0.0, end: float = 0.0):
            self.start, self.end = float(start), float(end)

        @property
        def duration(self) -> float:
            return self.end - self.start if self.end > self.start else 0.0

        @property
        def middle(self) -> float:
            return 0.5 * (self.start + self.end)

        def __iter__(self):
            yield self.start
            yield self.end

        def __bool__(self):
            return self.end - self.start > 1e-6

        def _key(self):
            return (self.start, self.end)

        def __eq__(self, other):
            return isinstance(other, Segment) and self._key() == other._key()

        def __lt__(self, other):
            return self._key() < other._key()

        def __hash__(self):
            return hash(self._key())

        def __repr__(self):
            return f"<Segment({self.start:g}, {self.end:g})>"

    class SlidingWindow:
        """Frame grid: frame i covers [start + i*step, start + i*step + duration)."""

        def __init__(self, duration: float = 0.030, step: float = 0.010, start: float = 0.0, end=None):
            self.duration, self.step, self.start, self.end = duration, step, start, end

        def __getitem__(self, i: int) -> Segment:
            s = self.start + i * self.step
            return Segment(s, s + self.duration)

        def closest_frame(self, t: float) -> int:
            return int(np.rint((t - self.start - 0.5 * self.duration) / self.step))

        def samples(self, from_duration: float, mode: str = "strict") -> int:
            if mode == "strict":
                return int(np.floor((from_duration - self.duration) / self.step)) + 1
            if mode == "loose":
                return int(np.floor((from_duration + self.duration) / self.step))
            if mode == "center":
                return int(np.rint(from_duration / self.step))
            raise ValueError(f"unknown mode '{mode}'")

        def crop_range(self, focus: Segment, mode: str = "loose", fixed: Optional[float] = None):
            """(first, last+1) frame indices of ``focus`` — may run out of the data bounds
            (pyannote.core SlidingWindow.crop(..., return_ranges=True) for one Segment)."""
            if mode == "loose":
                i = int(np.ceil((focus.start - self.duration - self.start) / self.step))
                j = int(np.floor((focus.end - self.start) / self.step))
            elif mode == "strict":
                i = int(np.ceil((focus.start - self.start) / self.step))
                j = int(np.floor((focus.end - self.duration - self.start) / self.step))
            elif mode == "center":
                i = self.closest_frame(focus.start)
                j = self.closest_frame(focus.end)
            else:
                raise ValueError(f"unknown mode '{mode}'")
            if fixed is None:
                return i, j + 1
            return i, i + self.samples(fixed, mode=mode)

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

        def crop(self, focus: Segment, mode: str = "loose", fixed: Optional[float] = None) -> np.ndarray:
            """Frames of ``focus`` as an array; with ``fixed`` the frame count is exact and frames
            requested outside the data repeat the first / last row (pyannote.core semantics)."""
            first, last = self.sliding_window.crop_range(focus, mode, fixed)
            n = self.data.shape[0]
            lo, hi = max(first, 0), min(last, n)
            body = self.data[lo:hi] if hi > lo else self.data[:0]
            if fixed is None:
                return body
            before = min(last, 0) - min(first, 0)
            after = max(last, n) - max(first, n)
            parts = []
            if before > 0:
                parts.append(np.repeat(self.data[:1], before, axis=0))
            parts.append(body)
            if after > 0:
                parts.append(np.repeat(self.data[n - 1:n], after, axis=0))
            return np.concatenate(parts, axis=0) if len(parts) > 1 else body

    class Timeline:
        def __init__(self, segments=None, uri=None):
            self.uri = uri
            self._segments = sorted(segments) if segments else []

        def add(self, segment: Segment):
            if segment and segment not in self._segments:
                self._segments.append(segment)
                self._segments.sort()
            return self

        def __iter__(self):
            return iter(self._segments)

        def __len__(self):
            return len(self._segments)

        def to_annotation(self, generator="string", modality=None) -> "Annotation":
            ann = Annotation(uri=self.uri, modality=modality)
            for n, seg in enumerate(self._segments):
                ann[seg, n] = next(generator) if hasattr(generator, "__next__") else f"{n}"
            return ann

    class Annotation:
        """Labelled speech turns: ``ann[segment, track] = label`` (the subset of
        pyannote.core.Annotation that the pipelines, RTTM I/O and the DER scorer use)."""

        def __init__(self, uri=None, modality=None):
            self.uri, self.modality = uri, modality
            self._tracks = {}  # (segment, track) -> label

        def __setitem__(self, key, label):
            segment, track = key if isinstance(key, tuple) else (key, "_")
            if segment:
                self._tracks[(segment, track)] = label

        def __len__(self):
            return len(self._tracks)

        def __bool__(self):
            return True

        def itertracks(self, yield_label: bool = False):
            for (segment, track), label in sorted(self._tracks.items(), key=lambda kv: (kv[0][0], str(kv[0][1]))):
                yield (segment, track, label) if yield_label else (segment, track)

        def labels(self):
            return sorted(set(self._tracks.values()), key=str)

        def get_timeline(self, copy: bool = True) -> Timeline:
            return Timeline({seg for seg, _ in self._tracks}, uri=self.uri)

        def update(self, other: "Annotation", copy: bool = False) -> "Annotation":
            self._tracks.update(other._tracks)
            return self

        def rename_labels(self, mapping=None, generator="string", copy=True) -> "Annotation":
            out = Annotation(self.uri, self.modality)
            for key, label in self._tracks.items():
                out._tracks[key] = mapping.get(label, label) if mapping else label
            return out

        def support(self, collar: float = 0.0) -> "Annotation":
            """Merge same-label turns that touch / overlap or whose gap is SHORTER than ``collar``
            (pyannote.core ``Timeline.support``: ``gap.duration < collar``, strict; a gap below the
            1e-6 s segment precision counts as empty)."""
            out = Annotation(self.uri, self.modality)
            by_label = {}
            for (seg, _), label in self._tracks.items():
                by_label.setdefault(label, []).append(seg)
            for label, segs in by_label.items():
                segs.sort()
                cur_s, cur_e, n = segs[0].start, segs[0].end, 0
                for seg in segs[1:]:
                    gap = seg.start - cur_e
                    if gap <= 1e-6 or gap < collar:
                        cur_e = max(cur_e, seg.end)
                    else:
                        out[Segment(cur_s, cur_e), f"{label}_{n}"] = label
                        cur_s, cur_e, n = seg.start, seg.end, n + 1
                out[Segment(cur_s, cur_e), f"{label}_{n}"] = label
            return out

        def to_rttm(self) -> str:
            uri = self.uri if self.uri else "<NA>"
            return "".join(
                f"SPEAKER {uri} 1 {seg.start:.3f} {seg.duration:.3f} <NA> <NA> {label} <NA> <NA>\n"
                for seg, _, label in self.itertracks(yield_label=True))

        def write_rttm(self, file):
            file.write(self.to_rttm())


def load_rttm(path) -> dict:
    """RTTM file -> {uri: Annotation} (pyannote.database.util.load_rttm surface)."""
    out = {}
    with open(path) as f:
        for n, line in enumerate(f):
            parts = line.split()
            if len(parts) < 8 or parts[0] != "SPEAKER":
                continue
            uri, start, dur, label = parts[1], float(parts[3]), float(parts[4]), parts[7]
            out.setdefault(uri, Annotation(uri=uri, modality="speaker"))[Segment(start, start + dur), n] = label
    return out


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
