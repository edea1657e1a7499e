"""``SpeakerDiarization`` pipeline (reference: ``/root/reference/src/diart/blocks/diarization.py``;
config :21-86, pipeline :89-234).  A ``blocks.Pipeline`` that ``StreamingInference`` /
``Benchmark`` / ``Optimizer`` can drive unchanged: same config fields and defaults, same
``__call__(Sequence[SlidingWindowFeature]) -> Sequence[(Annotation, SlidingWindowFeature)]``,
same asserts.  Segmentation, OSP, embedding and normalisation are HIP kernels; the K=3 redundant
embedding forward passes of the reference are one pass (``dz_emb_forward_multi``); clustering is
the C++ fp64 port; aggregation / binarisation are the vectorised host blocks of this package.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import models as m
from ..features import Annotation, Segment, SlidingWindow, SlidingWindowFeature
from . import base
from .aggregation import DelayedAggregation
from .clustering import OnlineSpeakerClustering
from .embedding import OverlapAwareSpeakerEmbedding
from .segmentation import SpeakerSegmentation
from .utils import Binarize


def _latency(latency, step, duration):
    if latency is None or latency == "min":
        return step
    if latency == "max":
        return duration
    return latency


class SpeakerDiarizationConfig(base.PipelineConfig):
    def __init__(self, segmentation: Optional[m.SegmentationModel] = None,
                 embedding: Optional[m.EmbeddingModel] = None, duration: float = 5, step: float = 0.5,
                 latency: Union[float, str, None] = None, tau_active: float = 0.6,
                 rho_update: float = 0.3, delta_new: float = 1, gamma: float = 3, beta: float = 10,
                 max_speakers: int = 20, normalize_embedding_weights: bool = False,
                 device: Optional[torch.device] = None, sample_rate: int = 16000, **kwargs):
        # the reference downloads pyannote/segmentation + pyannote/embedding here; offline the
        # models must be given (state dicts / checkpoints, see diart_amd.models)
        self.segmentation = segmentation or m.SegmentationModel.from_pyannote("pyannote/segmentation")
        self.embedding = embedding or m.EmbeddingModel.from_pyannote("pyannote/embedding")
        self._duration, self._sample_rate, self._step = duration, sample_rate, step
        self._latency = _latency(latency, step, duration)
        self.tau_active, self.rho_update, self.delta_new = tau_active, rho_update, delta_new
        self.gamma, self.beta, self.max_speakers = gamma, beta, max_speakers
        self.normalize_embedding_weights = normalize_embedding_weights
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


def shift_annotation(annotation: Annotation, shift: float) -> Annotation:
    out = Annotation(annotation.uri)
    for segment, track, speaker in annotation.itertracks(yield_label=True):
        out[Segment(segment.start + shift, segment.end + shift), track] = speaker
    return out


class SpeakerDiarization(base.Pipeline):
    def __init__(self, config: Optional[SpeakerDiarizationConfig] = None):
        self._config = SpeakerDiarizationConfig() if config is None else config
        c = self._config
        msg = f"Latency should be in the range [{c.step}, {c.duration}]"
        assert c.step <= c.latency <= c.duration, msg
        self.segmentation = SpeakerSegmentation(c.segmentation, c.device)
        self.embedding = OverlapAwareSpeakerEmbedding(c.embedding, c.gamma, c.beta, norm=1,
                                                      normalize_weights=c.normalize_embedding_weights,
                                                      device=c.device)
        self.pred_aggregation = DelayedAggregation(c.step, c.latency, strategy="hamming", cropping_mode="loose")
        self.audio_aggregation = DelayedAggregation(c.step, c.latency, strategy="first", cropping_mode="center")
        self.binarize = Binarize(c.tau_active)
        self.timestamp_shift = 0
        self.clustering = None
        self.chunk_buffer, self.pred_buffer = [], []
        self.reset()

    @staticmethod
    def get_config_class() -> type:
        return SpeakerDiarizationConfig

    @staticmethod
    def suggest_metric():
        from ..metrics import DiarizationErrorRate
        return DiarizationErrorRate(collar=0, skip_overlap=False)

    @staticmethod
    def hyper_parameters() -> Sequence[base.HyperParameter]:
        return [base.TauActive, base.RhoUpdate, base.DeltaNew]

    @property
    def config(self) -> SpeakerDiarizationConfig:
        return self._config

    def set_timestamp_shift(self, shift: float):
        self.timestamp_shift = shift

    def reset(self):
        self.set_timestamp_shift(0)
        c = self.config
        self.clustering = OnlineSpeakerClustering(c.tau_active, c.rho_update, c.delta_new, "cosine", c.max_speakers)
        self.chunk_buffer, self.pred_buffer = [], []

    def __call__(self, waveforms: Sequence[SlidingWindowFeature]) -> Sequence[Tuple[Annotation, SlidingWindowFeature]]:
        batch_size = len(waveforms)
        assert batch_size >= 1, "Pipeline expected at least 1 input"
        batch = torch.stack([torch.from_numpy(w.data) for w in waveforms])   # (batch, samples, channels)
        expected = int(np.rint(self.config.duration * self.config.sample_rate))
        assert batch.shape[1] == expected, f"Expected {expected} samples per chunk, but got {batch.shape[1]}"

        segmentations = self.segmentation(batch)                 # (batch, frames, speakers), host
        embeddings = self.embedding(batch, segmentations)        # (batch, speakers, emb_dim), host
        seg_resolution = waveforms[0].extent.duration / segmentations.shape[1]

        outputs = []
        for wav, seg, emb in zip(waveforms, segmentations, embeddings):
            sw = SlidingWindow(start=wav.extent.start, duration=seg_resolution, step=seg_resolution)
            seg = SlidingWindowFeature(seg.cpu().numpy(), sw)
            permuted_seg = self.clustering(seg, emb)
            self.chunk_buffer.append(wav)
            self.pred_buffer.append(permuted_seg)
            agg_waveform = self.audio_aggregation(self.chunk_buffer)
            agg_prediction = self.binarize(self.pred_aggregation(self.pred_buffer))
            if self.timestamp_shift != 0:
                agg_prediction = shift_annotation(agg_prediction, self.timestamp_shift)
            outputs.append((agg_prediction, agg_waveform))
            if len(self.chunk_buffer) == self.pred_aggregation.num_overlapping_windows:
                self.chunk_buffer = self.chunk_buffer[1:]
                self.pred_buffer = self.pred_buffer[1:]
        return outputs
