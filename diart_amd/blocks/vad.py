"""``VoiceActivityDetection`` pipeline (reference: ``/root/reference/src/diart/blocks/vad.py``;
config :26-73, pipeline :76-191): segmentation -> max over speakers -> aggregation -> binarise.
The hot-path lines are :136-148 (segmentation + ``torch.max(..., dim=-1, keepdim=True)``)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import models as m
from ..features import Annotation, Segment, SlidingWindow, SlidingWindowFeature, Timeline
from . import base
from .aggregation import DelayedAggregation
from .diarization import _latency
from .segmentation import SpeakerSegmentation
from .utils import Binarize, windows_batch


def _repeat_label(label):
    while True:
        yield label


class VoiceActivityDetectionConfig(base.PipelineConfig):
    def __init__(self, segmentation: Optional[m.SegmentationModel] = None, duration: float = 5,
                 step: float = 0.5, latency: Union[float, str, None] = None, tau_active: float = 0.6,
                 device: Optional[torch.device] = None, sample_rate: int = 16000, **kwargs):
        self.segmentation = segmentation or m.SegmentationModel.from_pyannote("pyannote/segmentation")
        self._duration, self._step, self._sample_rate = duration, step, sample_rate
        self._latency = _latency(latency, step, duration)
        self.tau_active = tau_active
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")

    @property
    def duration(self) -> float:
        return self._duration

    @property
    def step(self) -> float:
        return self._step

    @property
    def latency(self) -> float:
        return self._latency

    @property
    def sample_rate(self) -> int:
        return self._sample_rate


class VoiceActivityDetection(base.Pipeline):
    def __init__(self, config: Optional[VoiceActivityDetectionConfig] = None):
        self._config = VoiceActivityDetectionConfig() if config is None else config
        c = self._config
        msg = f"Latency should be in the range [{c.step}, {c.duration}]"
        assert c.step <= c.latency <= c.duration, msg
        self.segmentation = SpeakerSegmentation(c.segmentation, c.device)
        self.pred_aggregation = DelayedAggregation(c.step, c.latency, strategy="hamming", cropping_mode="loose")
        self.audio_aggregation = DelayedAggregation(c.step, c.latency, strategy="first", cropping_mode="center")
        self.binarize = Binarize(c.tau_active)
        self.timestamp_shift = 0
        self.chunk_buffer, self.pred_buffer = [], []

    @staticmethod
    def get_config_class() -> type:
        return VoiceActivityDetectionConfig

    @staticmethod
    def suggest_metric():
        from ..metrics import DetectionErrorRate
        return DetectionErrorRate(collar=0, skip_overlap=False)

    @staticmethod
    def hyper_parameters() -> Sequence[base.HyperParameter]:
        return [base.TauActive]

    @property
    def config(self) -> VoiceActivityDetectionConfig:
        return self._config

    def reset(self):
        self.set_timestamp_shift(0)
        self.chunk_buffer, self.pred_buffer = [], []

    def set_timestamp_shift(self, shift: float):
        self.timestamp_shift = shift

    def __call__(self, waveforms: Sequence[SlidingWindowFeature]) -> Sequence[Tuple[Annotation, SlidingWindowFeature]]:
        assert len(waveforms) >= 1, "Pipeline expected at least 1 input"
        expected = int(np.rint(self.config.duration * self.config.sample_rate))
        got = waveforms[0].data.shape[0]
        assert all(w.data.shape[0] == got for w in waveforms), "chunks of different lengths in one batch"
        assert got == expected, f"Expected {expected} samples per chunk, but got {got}"
        batch = windows_batch(waveforms, self.config.device)
        return self.finalise(waveforms, self.segmentation(batch))

    def finalise(self, waveforms: Sequence[SlidingWindowFeature], segmentations: torch.Tensor):
        """The host half of ``__call__`` (reference vad.py:146-191) for given segmentation scores."""
        voice_detection = torch.max(segmentations, dim=-1, keepdim=True)[0]   # (batch, frames, 1)
        seg_resolution = waveforms[0].extent.duration / segmentations.shape[1]
        outputs = []
        for wav, vad in zip(waveforms, voice_detection):
            sw = SlidingWindow(start=wav.extent.start, duration=seg_resolution, step=seg_resolution)
            vad = SlidingWindowFeature(vad.cpu().numpy(), sw)
            self.chunk_buffer.append(wav)
            self.pred_buffer.append(vad)
            agg_waveform = self.audio_aggregation(self.chunk_buffer)
            timeline = self.binarize(self.pred_aggregation(self.pred_buffer)).get_timeline(copy=False)
            if self.timestamp_shift != 0:
                shifted = Timeline(uri=timeline.uri)
                for segment in timeline:
                    shifted.add(Segment(segment.start + self.timestamp_shift, segment.end + self.timestamp_shift))
                timeline = shifted
            outputs.append((timeline.to_annotation(_repeat_label("speech")), agg_waveform))
            if len(self.chunk_buffer) == self.pred_aggregation.num_overlapping_windows:
                self.chunk_buffer = self.chunk_buffer[1:]
                self.pred_buffer = self.pred_buffer[1:]
        return outputs
