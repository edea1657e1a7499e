"""``Binarize``: frame scores -> speech turns (reference:
``/root/reference/src/diart/blocks/utils.py:11-59``).  Same turns, found per speaker from the
rising / falling edges of ``score > threshold`` instead of a Python loop over frames; turn
boundaries are frame middles, a turn still open at the last frame closes at the middle of the
(virtual) frame after it, labels are ``speaker{index}`` and the track is the speaker index."""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..features import Annotation, Segment, SlidingWindowFeature


class Binarize:
    def __init__(self, threshold: float, uri: Optional[str] = None):
        self.uri, self.threshold = uri, threshold

    def __call__(self, segmentation: SlidingWindowFeature) -> Annotation:
        num_frames, num_speakers = segmentation.data.shape
        ts = segmentation.sliding_window
        active = np.zeros((num_frames + 2, num_speakers), dtype=np.int8)
        active[1:-1] = segmentation.data > self.threshold       # strict, utils.py:45
        edges = np.diff(active, axis=0)                          # (num_frames + 1, speakers)
        annotation = Annotation(uri=self.uri, modality="speech")
        for spk in np.nonzero(edges.any(axis=0))[0]:
            onsets = np.nonzero(edges[:, spk] == 1)[0]           # first active frame
            offsets = np.nonzero(edges[:, spk] == -1)[0]         # first inactive frame after it
            for a, b in zip(onsets, offsets):
                annotation[Segment(ts[int(a)].middle, ts[int(b)].middle), int(spk)] = f"speaker{spk}"
        return annotation
