"""Minimal stand-in for the two ``pyannote.core`` symbols the reference's hot-path files touch,
plus a by-path loader for those files.  TEST INFRASTRUCTURE, used only in the build container
(``/root/reference`` does not exist on the GPU box) by ``tests/golden/make_golden.py`` to produce
the committed fixtures, i.e. to pin the oracle against the reference's OWN code:

* ``/root/reference/src/diart/functional.py``, ``mapping.py``, ``features.py``
* ``/root/reference/src/diart/blocks/clustering.py``, ``blocks/embedding.py``, ``blocks/segmentation.py``

``pyannote.core`` itself is not installed here.  Surface provided:
``SlidingWindow(start, duration, step)``, ``SlidingWindowFeature(data, sliding_window)`` with
``.data / .sliding_window / .extent / __getitem__``, ``Segment``, and
``pyannote.core.utils.distance.cdist`` == ``scipy.spatial.distance.cdist`` for "cosine".
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np


class Segment:
    def __init__(self, start: float, end: float):
        self.start, self.end = start, end

    @property
    def duration(self):
        return self.end - self.start

    @property
    def middle(self):
        return 0.5 * (self.start + self.end)


class SlidingWindow:
    def __init__(self, duration: float = 0.03, step: float = 0.01, start: float = 0.0, end=None):
        self.duration, self.step, self.start, self.end = duration, step, start, end

    def __getitem__(self, i: int) -> Segment:
        s = self.start + i * self.step
        return Segment(s, s + self.duration)


    # --- pyannote.core.SlidingWindow.closest_frame / samples / crop (one Segment, ranges) -----
    def closest_frame(self, t: float) -> int:
        return int(np.rint((t - self.start - .5 * self.duration) / self.step))

    def samples(self, from_duration: float, mode: str = "strict") -> int:
        if mode == "strict":
            return int(np.floor((from_duration - self.duration) / self.step)) + 1
        elif mode == "loose":
            return int(np.floor((from_duration + self.duration) / self.step))
        elif mode == "center":
            return int(np.rint((from_duration / self.step)))

    def crop(self, focus: "Segment", mode: str = "loose", fixed=None, return_ranges: bool = True):
        if mode == "loose":
            i = int(np.ceil((focus.start - self.duration - self.start) / self.step))
            if fixed is None:
                j = int(np.floor((focus.end - self.start) / self.step))
                rng = (i, j + 1)
            else:
                rng = (i, i + self.samples(fixed, mode="loose"))
        elif mode == "strict":
            i = int(np.ceil((focus.start - self.start) / self.step))
            if fixed is None:
                j = int(np.floor((focus.end - self.duration - self.start) / self.step))
                rng = (i, j + 1)
            else:
                rng = (i, i + self.samples(fixed, mode="strict"))
        elif mode == "center":
            i = self.closest_frame(focus.start)
            if fixed is None:
                rng = (i, self.closest_frame(focus.end) + 1)
            else:
                rng = (i, i + self.samples(fixed, mode="center"))
        else:
            raise ValueError(mode)
        return [rng]


class SlidingWindowFeature:
    def __init__(self, data: np.ndarray, sliding_window: SlidingWindow):
        self.data, self.sliding_window = data, sliding_window

    def __getitem__(self, i):
        return self.data[i]

    def crop(self, focus: Segment, mode: str = "loose", fixed=None, return_data: bool = True):
        """pyannote.core.SlidingWindowFeature.crop for a Segment focus (returns the ndarray)."""
        ranges = self.sliding_window.crop(focus, mode=mode, fixed=fixed, return_ranges=True)
        n_samples = self.data.shape[0]
        n_dimensions = len(self.data.shape) - 1
        clipped_ranges, repeat_first, repeat_last = [], 0, 0
        for start, end in ranges:
            repeat_first += min(end, 0) - min(start, 0)
            repeat_last += max(end, n_samples) - max(start, n_samples)
            if end < 0 or start >= n_samples:
                continue
            clipped_ranges += [[max(start, 0), min(end, n_samples)]]
        if clipped_ranges:
            data = np.vstack([self.data[start:end, :] for start, end in clipped_ranges])
        else:
            data = np.empty((0,) + self.data.shape[1:])
        if fixed is not None:
            data = np.vstack([np.tile(self.data[0], (repeat_first,) + (1,) * n_dimensions), data,
                              np.tile(self.data[n_samples - 1], (repeat_last,) + (1,) * n_dimensions)])
        return data

    def __len__(self):
        return self.data.shape[0]

    @property
    def extent(self) -> Segment:
        sw = self.sliding_window
        n = self.data.shape[0]
        return Segment(sw.start, sw.start + (n - 1) * sw.step + sw.duration)


class Annotation:
    """``annotation[segment, track] = label`` + ``itertracks`` (what Binarize / the pipelines use)."""

    def __init__(self, uri=None, modality=None):
        self.uri, self.modality = uri, modality
        self.tracks = []

    def __setitem__(self, key, label):
        segment, track = key
        if segment.end - segment.start > 1e-6:      # pyannote ignores empty segments
            self.tracks.append((segment, track, label))

    def itertracks(self, yield_label=False):
        for seg, track, label in sorted(self.tracks, key=lambda t: (t[0].start, t[0].end, str(t[1]))):
            yield (seg, track, label) if yield_label else (seg, track)


class Timeline:
    """Sorted set of segments (what VoiceActivityDetection's tail touches: vad.py:169-183)."""

    def __init__(self, segments=None, uri=None):
        self.uri = uri
        self.segments = []
        for seg in segments or ():
            self.add(seg)

    def add(self, segment):
        if segment.end - segment.start > 1e-6 and not any(
                s.start == segment.start and s.end == segment.end for s in self.segments):
            self.segments.append(segment)
        return self

    def __iter__(self):
        return iter(sorted(self.segments, key=lambda s: (s.start, s.end)))

    def __len__(self):
        return len(self.segments)

    def to_annotation(self, generator="string", modality=None):
        ann = Annotation(uri=self.uri, modality=modality)
        for n, seg in enumerate(self):
            ann[seg, "_"] = next(generator) if hasattr(generator, "__next__") else str(n)
        return ann


def _annotation_get_timeline(self, copy=True):
    return Timeline([seg for seg, _, _ in self.tracks], uri=self.uri)


Annotation.get_timeline = _annotation_get_timeline


def install() -> None:
    """Register the stand-in as ``pyannote.core`` (only if the real one is absent)."""
    if "pyannote.core" in sys.modules:
        return
    from scipy.spatial.distance import cdist
    pkg = types.ModuleType("pyannote")
    pkg.__path__ = []
    core = types.ModuleType("pyannote.core")
    core.__path__ = []
    core.SlidingWindow, core.SlidingWindowFeature, core.Segment = SlidingWindow, SlidingWindowFeature, Segment
    core.Annotation, core.Timeline = Annotation, Timeline
    core.notebook = types.SimpleNamespace()      # diart/utils.py imports the plotting helper object
    if "torchaudio" not in sys.modules:   # blocks/utils.py imports torchaudio.transforms for Resample
        ta = types.ModuleType("torchaudio")
        ta.__path__ = []
        tat = types.ModuleType("torchaudio.transforms")
        taf = types.ModuleType("torchaudio.functional")   # audio.py imports it for resampling (never called here)
        taf.resample = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torchaudio is not installed"))
        ta.set_audio_backend = lambda name: None          # audio.py calls it at import
        ta.transforms, ta.functional = tat, taf
        sys.modules.update({"torchaudio": ta, "torchaudio.transforms": tat, "torchaudio.functional": taf})
    utils = types.ModuleType("pyannote.core.utils")
    utils.__path__ = []
    dist = types.ModuleType("pyannote.core.utils.distance")
    dist.cdist = cdist
    sys.modules.update({"pyannote": pkg, "pyannote.core": core, "pyannote.core.utils": utils,
                        "pyannote.core.utils.distance": dist})


def load_reference(root: str = "/root/reference/src/diart"):
    """Import the reference's hot-path modules by path under the package name ``diart_ref``
    WITHOUT running its ``__init__`` files (they pull in rx / pyannote.audio / torchaudio).
    Returns a namespace with .functional .mapping .features .models .clustering .embedding
    .segmentation."""
    install()
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree
    rootp = Path(root)
    if not rootp.exists():
        raise FileNotFoundError(f"{root} is not available (only present in the build container)")
    for name, path in (("diart_ref", rootp), ("diart_ref.blocks", rootp / "blocks")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [str(path)]
            sys.modules[name] = m

    def _load(name: str, rel: str):
        full = f"diart_ref.{name}"
        if full in sys.modules:
            return sys.modules[full]
        spec = importlib.util.spec_from_file_location(full, rootp / rel)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        return mod

    ns = types.SimpleNamespace()
    ns.functional = _load("functional", "functional.py")
    ns.mapping = _load("mapping", "mapping.py")
    ns.features = _load("features", "features.py")
    ns.models = _load("models", "models.py")
    ns.clustering = _load("blocks.clustering", "blocks/clustering.py")
    ns.embedding = _load("blocks.embedding", "blocks/embedding.py")
    ns.segmentation = _load("blocks.segmentation", "blocks/segmentation.py")
    ns.aggregation = _load("blocks.aggregation", "blocks/aggregation.py")
    ns.blocks_utils = _load("blocks.utils", "blocks/utils.py")
    return ns


def load_reference_pipelines(root: str = "/root/reference/src/diart"):
    """``load_reference`` plus the reference's PIPELINE classes — ``blocks/base.py``,
    ``blocks/diarization.py``, ``blocks/vad.py`` (and the ``utils.py`` / ``audio.py`` / ``progress.py``
    they import) — by path, unmodified.  ``pyannote.metrics`` is absent: the two metric classes the
    pipelines only hand out from ``suggest_metric`` are replaced by name-carrying placeholders."""
    ns = load_reference(root)
    rootp = Path(root)
    if "pyannote.metrics" not in sys.modules:
        pm = types.ModuleType("pyannote.metrics")
        pm.__path__ = []
        pmb = types.ModuleType("pyannote.metrics.base")
        pmb.BaseMetric = type("BaseMetric", (), {})
        pmd = types.ModuleType("pyannote.metrics.diarization")
        pmd.DiarizationErrorRate = type("DiarizationErrorRate", (pmb.BaseMetric,), {"__init__": lambda self, **kw: setattr(self, "kw", kw)})
        pmt = types.ModuleType("pyannote.metrics.detection")
        pmt.DetectionErrorRate = type("DetectionErrorRate", (pmb.BaseMetric,), {"__init__": lambda self, **kw: setattr(self, "kw", kw)})
        sys.modules.update({"pyannote.metrics": pm, "pyannote.metrics.base": pmb,
                            "pyannote.metrics.diarization": pmd, "pyannote.metrics.detection": pmt})

    def _load(name: str, rel: str):
        full = f"diart_ref.{name}"
        if full in sys.modules:
            return sys.modules[full]
        spec = importlib.util.spec_from_file_location(full, rootp / rel)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        return mod

    _load("progress", "progress.py")
    ns.utils = _load("utils", "utils.py")
    _load("audio", "audio.py")
    ns.base = _load("blocks.base", "blocks/base.py")
    ns.diarization = _load("blocks.diarization", "blocks/diarization.py")
    ns.vad = _load("blocks.vad", "blocks/vad.py")
    return ns
