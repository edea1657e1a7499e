"""``SpeakerSegmentation`` block (reference: ``/root/reference/src/diart/blocks/segmentation.py``).

waveform ``(samples, channels)`` or ``(batch, samples, channels)`` as SlidingWindowFeature /
ndarray / Tensor -> speaker activations ``(batch, frames, speakers)`` of the same kind, on the
host, exactly like the reference block; the forward pass itself is ``dz_seg_forward``.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..features import TemporalFeatureFormatter, TemporalFeatures
from ..models import SegmentationModel


def _range_check(device) -> None:
    """After the synchronising ``.cpu()``: an f16x3 operand beyond +-65504 raises (it was clamped)."""
    if getattr(device, "type", None) == "cuda":
        from .. import _lib
        _lib.range_check(device.index if device.index is not None else torch.cuda.current_device())


class SpeakerSegmentation:
    def __init__(self, model: SegmentationModel, device: Optional[torch.device] = None):
        self.model = model
        self.model.eval()
        self.device = device if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.model.to(self.device)
        self.formatter = TemporalFeatureFormatter()

    @staticmethod
    def from_pretrained(model, use_hf_token=True, device: Optional[torch.device] = None):
        return SpeakerSegmentation(SegmentationModel.from_pretrained(model, use_hf_token), device)

    def forward_device(self, waveform: TemporalFeatures) -> torch.Tensor:
        """``__call__`` without the trip to the host: (batch, frames, speakers) on the model's device, nothing
        synchronised — for a pipeline that feeds the embedding block from it and copies both results once."""
        wave = self.formatter.cast(waveform)
        if wave.shape[2] != 1:
            raise ValueError(f"expected mono audio, got {wave.shape[2]} channels")
        with torch.no_grad():
            return self.model(wave.transpose(1, 2).to(self.device))

    def __call__(self, waveform: TemporalFeatures) -> TemporalFeatures:
        wave = self.formatter.cast(waveform)            # (batch, samples, channels) float32
        if wave.shape[2] != 1:
            raise ValueError(f"expected mono audio, got {wave.shape[2]} channels")
        # (b, s, 1) -> (b, 1, s) is a pure view for mono audio: no transpose kernel, no copy
        rows = wave.transpose(1, 2)
        with torch.no_grad():
            out = self.model(rows.to(self.device)).cpu()
        _range_check(self.device)
        return self.formatter.restore_type(out)
