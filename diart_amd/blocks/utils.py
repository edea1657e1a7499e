"""``Binarize``: frame scores -> speech turns (reference:
``/root/reference/src/diart/blocks/utils.py:11-59``).  Same turns, found per speaker from the
rising / falling edges of ``score > threshold`` instead of a Python loop over frames; turn
boundaries are frame middles, a turn still open at the last frame closes at the middle of the
(virtual) frame after it, labels are ``speaker{index}`` and the track is the speaker index."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from ..features import Annotation, Segment, SlidingWindowFeature


def windows_batch(waveforms: Sequence[SlidingWindowFeature], device=None) -> torch.Tensor:
    """``(batch, samples, channels)`` float tensor of the chunks' audio ON ``device`` — what the reference builds
    with ``torch.stack`` (/root/reference/src/diart/blocks/diarization.py:175) and each block then uploads on its
    own (segmentation.py:47, embedding.py:52: the same 320 KB per chunk twice).  Here the batch is uploaded ONCE
    and both blocks get the device tensor.  When the chunks are overlapping views of one host buffer at a
    constant hop — what a rolling window over a file or a stream is: 32 windows of 5 s at a 500 ms step are 20.5 s
    of audio, not 160 s — only the span they cover crosses PCIe and the batch is a strided view of it on the device
    (the kernels read rows at any stride).  Values are identical either way."""
    datas = [w.data for w in waveforms]
    dev = torch.device("cpu") if device is None else torch.device(device)
    span = _common_span(datas) if dev.type == "cuda" else None
    if span is not None:
        flat, hop, n = span
        import warnings
        with warnings.catch_warnings():          # (a read-only source buffer: nothing here writes through the view)
            warnings.simplefilter("ignore", UserWarning)
            d = torch.from_numpy(flat).to(dev)
        return d.as_strided((len(datas), n, 1), (hop, 1, 1))
    return torch.stack([torch.from_numpy(x) for x in datas]).to(dev)


def _common_span(datas):
    """(1-D host view covering every window, hop in samples, samples per window) when the windows are mono float32
    views of ONE buffer at a constant, 16-byte aligned, non-negative hop; else None."""
    if len(datas) < 2:
        return None
    first = datas[0]
    if not isinstance(first, np.ndarray) or first.dtype != np.float32 or first.ndim != 2 or first.shape[1] != 1:
        return None
    n = first.shape[0]

    def owner(a):
        while isinstance(getattr(a, "base", None), np.ndarray):
            a = a.base
        return a

    root = owner(first)
    if root is first or not isinstance(root, np.ndarray):
        return None
    addrs = []
    for x in datas:
        if not isinstance(x, np.ndarray) or x.dtype != np.float32 or x.shape != (n, 1) or x.strides[0] != 4 or owner(x) is not root:
            return None
        addrs.append(x.__array_interface__["data"][0])
    hop_b = addrs[1] - addrs[0]
    if hop_b < 0 or hop_b % 16 or any(b - a != hop_b for a, b in zip(addrs, addrs[1:])):
        return None
    if hop_b >= 4 * n:          # windows that do not overlap (far apart in one large or mmap'd buffer): the span would
        return None             # move MORE than the stacked batch does
    lo, hi = root.__array_interface__["data"][0], root.__array_interface__["data"][0] + root.nbytes
    if not root.flags["C_CONTIGUOUS"] or addrs[0] < lo or addrs[-1] + 4 * n > hi:
        return None
    total = (addrs[-1] - addrs[0]) // 4 + n
    flat = np.lib.stride_tricks.as_strided(first[:, 0], shape=(total,), strides=(4,))
    return flat, hop_b // 4, n


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
