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
from .. import _lib
from .aggregation import BatchedOutputTail, DelayedAggregation
from .clustering import OnlineSpeakerClustering
from .embedding import OverlapAwareSpeakerEmbedding
from .segmentation import SpeakerSegmentation
from .utils import Binarize, windows_batch


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
        self._tail = None
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
        if self._tail is not None:
            self._tail.reset()

    def __call__(self, waveforms: Sequence[SlidingWindowFeature]) -> Sequence[Tuple[Annotation, SlidingWindowFeature]]:
        batch_size = len(waveforms)
        assert batch_size >= 1, "Pipeline expected at least 1 input"
        expected = int(np.rint(self.config.duration * self.config.sample_rate))
        got = waveforms[0].data.shape[0]
        assert all(w.data.shape[0] == got for w in waveforms), "chunks of different lengths in one batch"
        assert got == expected, f"Expected {expected} samples per chunk, but got {got}"
        # (batch, samples, channels) on the device, uploaded once for both blocks (blocks/utils.py)
        batch = windows_batch(waveforms, self.config.device)

        if batch.is_cuda:
            # the segmentation stays on the device for the embedding block (OSP weights / masks are computed there
            # anyway); one synchronising copy per result instead of host round trips between the two blocks
            seg_dev = self.segmentation.forward_device(batch)
            embeddings = self.embedding(batch, seg_dev)          # (batch, speakers, emb_dim), host
            segmentations = seg_dev.cpu()                        # (batch, frames, speakers), host
        else:
            segmentations = self.segmentation(batch)
            embeddings = self.embedding(batch, segmentations)
        return self.finalise(waveforms, segmentations, embeddings)

    def finalise(self, waveforms: Sequence[SlidingWindowFeature], segmentations: torch.Tensor,
                 embeddings: torch.Tensor) -> Sequence[Tuple[Annotation, SlidingWindowFeature]]:
        """The host half of ``__call__`` (reference diarization.py:190-232): given the chunks and the
        segmentation / embeddings the models produced for them, step this stream's clustering,
        aggregation and binarisation once per chunk, in order.  Split out so that it can be driven
        with the outputs of the REFERENCE's own blocks (tests/test_reference_pipeline.py)."""
        seg_resolution = waveforms[0].extent.duration / segmentations.shape[1]

        # Clustering -> DelayedAggregation -> Binarize of the B consecutive chunks, in order, in ONE
        # C++ call (dz_file_step_batch on this stream's own clustering / aggregation state): what the
        # reference does chunk by chunk in Python (diarization.py:193-232).  The C++ tail is
        # bit-identical to the Python blocks of this package (tests/test_tail.py), which remain the
        # stand-alone DelayedAggregation / Binarize of the API.
        seg_np = np.ascontiguousarray(segmentations.detach().cpu().numpy(), dtype=np.float32)
        emb_np = np.ascontiguousarray(embeddings.detach().cpu().numpy(), dtype=np.float32)
        if emb_np.ndim == 2:
            emb_np = emb_np[None]
        B, F, K = seg_np.shape
        if emb_np.shape[:2] != (B, K):
            raise ValueError(f"expected embeddings (batch, speakers, dim) for segmentation {seg_np.shape}, "
                             f"got {emb_np.shape}")
        tail = self._output_tail(F)
        starts = np.array([w.extent.start for w in waveforms], dtype=np.float64)
        turns = np.empty((B, tail.max_turns, 3), dtype=np.float64)
        nturns = np.zeros(B, dtype=np.int32)
        lib = _lib.load()
        row0, count = np.zeros(1, dtype=np.int32), np.array([B], dtype=np.int32)   # (kept alive across the call)
        rc = lib.dz_file_step_batch(
            (_lib.vp * 1)(self.clustering._h), (_lib.vp * 1)(tail._hs[0]), 1,
            row0.ctypes.data, count.ctypes.data,
            seg_np.ctypes.data, F, K, emb_np.ctypes.data, emb_np.shape[2], self.config.max_speakers,
            starts.ctypes.data, float(seg_resolution), turns.ctypes.data, tail.max_turns, nturns.ctypes.data,
            None, 1)
        if rc == 3:   # the reference raises here too (assert / scipy ValueError)
            raise AssertionError(lib.dz_last_error().decode())
        _lib.check(rc, "dz_file_step_batch")

        outputs = []
        for i, wav in enumerate(waveforms):
            self.chunk_buffer.append(wav)
            agg_waveform = self.audio_aggregation(self.chunk_buffer)
            agg_prediction = BatchedOutputTail.annotation(turns[i], int(nturns[i]),
                                                          shift=self.timestamp_shift if self.timestamp_shift != 0 else 0.0)
            outputs.append((agg_prediction, agg_waveform))
            if len(self.chunk_buffer) == self.pred_aggregation.num_overlapping_windows:
                self.chunk_buffer = self.chunk_buffer[1:]
        return outputs

    def _output_tail(self, frames: int) -> "BatchedOutputTail":
        c = self.config
        if self._tail is None or self._tail.F != frames:
            self._tail = BatchedOutputTail(1, frames, c.max_speakers, c.step, c.latency, c.tau_active, num_threads=1)
        return self._tail
