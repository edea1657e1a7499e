"""Embedding blocks (reference: ``/root/reference/src/diart/blocks/embedding.py``).

``SpeakerEmbedding`` (:11-68), ``OverlappedSpeechPenalty`` (:71-107), ``EmbeddingNormalization``
(:110-120) and ``OverlapAwareSpeakerEmbedding`` (:123-178) keep their signatures and results.
Differences in HOW, not WHAT:

* the reference materialises K copies of every waveform and runs the whole network on each
  (``inputs.repeat(1, num_speakers, 1)``, :57); when the wrapped model offers
  ``forward_multi`` (``HipEmbedding`` does) the waveform is passed once and the K weight
  tracks are pooled from one set of frame features — identical output, 1/K of the FLOPs;
  any other model (e.g. a user's custom callable) takes the reference's repeat path;
* OSP and the final L2 normalisation are HIP kernels; in ``OverlapAwareSpeakerEmbedding`` the
  weights go from the OSP kernel to the pooling kernel without leaving the GPU.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from .. import functional as F
from ..features import TemporalFeatureFormatter, TemporalFeatures
from ..models import EmbeddingModel
from .segmentation import _range_check


def _default_device(device):
    return device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")


class SpeakerEmbedding:
    def __init__(self, model: EmbeddingModel, device: Optional[torch.device] = None):
        self.model = model
        self.model.eval()
        self.device = _default_device(device)
        self.model.to(self.device)
        self.waveform_formatter = TemporalFeatureFormatter()
        self.weights_formatter = TemporalFeatureFormatter()

    @staticmethod
    def from_pretrained(model, use_hf_token=True, device: Optional[torch.device] = None):
        return SpeakerEmbedding(EmbeddingModel.from_pretrained(model, use_hf_token), device)

    def _embed(self, wave: torch.Tensor, weights: Optional[torch.Tensor],
               speaker_major: bool = False, normalize: bool = False) -> torch.Tensor:
        """wave (b, s, 1) on any device; weights (b, f, k) — or (b, k, f) if speaker_major —
        -> (b, k, d) / (b, d) on the GPU (not squeezed)."""
        rows = wave.to(self.device).transpose(1, 2)                 # (b, 1, s) view
        if weights is None:
            return self.model(rows)
        weights = weights.to(self.device)
        inner = getattr(self.model, "model", None)
        if hasattr(inner, "forward_multi"):
            wk = weights if speaker_major else weights.transpose(1, 2)
            return inner.forward_multi(rows, wk.contiguous(), normalize=normalize)
        # generic model: the reference's (batch spk) repetition, embedding.py:56-65
        if speaker_major:
            weights = weights.transpose(1, 2)
        b, _, k = weights.shape
        rep = rows.repeat(1, k, 1).reshape(b * k, 1, rows.shape[-1])
        wrows = weights.permute(0, 2, 1).reshape(b * k, -1)
        out = self.model(rep, wrows).reshape(b, k, -1)
        return F.normalize_embeddings(out) if normalize else out

    def __call__(self, waveform: TemporalFeatures, weights: Optional[TemporalFeatures] = None) -> torch.Tensor:
        with torch.no_grad():
            wave = self.waveform_formatter.cast(waveform)
            w = self.weights_formatter.cast(weights) if weights is not None else None
            out = self._embed(wave, w).squeeze().cpu()              # embedding.py:68
            _range_check(self.device)
            return out


class OverlappedSpeechPenalty:
    """Down-weights overlapped and low-confidence frames (paper Eq. 2)."""

    def __init__(self, gamma: float = 3, beta: float = 10, normalize: bool = False):
        self.gamma, self.beta, self.normalize = gamma, beta, normalize
        self.formatter = TemporalFeatureFormatter()

    def __call__(self, segmentation: TemporalFeatures) -> TemporalFeatures:
        seg = self.formatter.cast(segmentation)
        weights = F.overlapped_speech_penalty(seg, self.gamma, self.beta, self.normalize)
        return self.formatter.restore_type(weights)


class EmbeddingNormalization:
    def __init__(self, norm: Union[float, torch.Tensor] = 1):
        self.norm = norm
        if isinstance(self.norm, torch.Tensor) and self.norm.ndim == 2:
            self.norm = self.norm.unsqueeze(0)

    def __call__(self, embeddings: torch.Tensor) -> torch.Tensor:
        return F.normalize_embeddings(embeddings, self.norm)


class OverlapAwareSpeakerEmbedding:
    """normalize(embedding(waveform, osp(segmentation))) with the whole chain on the GPU."""

    def __init__(self, model: EmbeddingModel, gamma: float = 3, beta: float = 10,
                 norm: Union[float, torch.Tensor] = 1, normalize_weights: bool = False,
                 device: Optional[torch.device] = None):
        self.embedding = SpeakerEmbedding(model, device)
        self.osp = OverlappedSpeechPenalty(gamma, beta, normalize_weights)
        self.normalize = EmbeddingNormalization(norm)

    @staticmethod
    def from_pretrained(model, gamma: float = 3, beta: float = 10, norm=1, use_hf_token=True,
                        normalize_weights: bool = False, device: Optional[torch.device] = None):
        model = EmbeddingModel.from_pretrained(model, use_hf_token)
        return OverlapAwareSpeakerEmbedding(model, gamma, beta, norm, normalize_weights, device)

    def __call__(self, waveform: TemporalFeatures, segmentation: TemporalFeatures) -> torch.Tensor:
        emb = self.embedding
        with torch.no_grad():
            wave = emb.waveform_formatter.cast(waveform)
            seg = emb.weights_formatter.cast(segmentation).to(emb.device)
            weights = F.overlapped_speech_penalty(seg, self.osp.gamma, self.osp.beta,
                                                  self.osp.normalize, speaker_major=True)
            unit = isinstance(self.normalize.norm, (int, float)) and self.normalize.norm == 1
            out = emb._embed(wave, weights, speaker_major=True, normalize=unit)
            if not unit:
                out = self.normalize(out)
            out = out.squeeze()                                     # embedding.py:68
            if out.ndim == 2:                                       # functional.py:20-21
                out = out.unsqueeze(0)
            out = out.cpu()
            _range_check(emb.device)
            return out
