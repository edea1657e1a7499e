"""Tensor functions of the hot path, computed by HIP kernels.

Same names and results as ``/root/reference/src/diart/functional.py`` (:6-13
``overlapped_speech_penalty``, :16-27 ``normalize_embeddings``).  Inputs may live on the host
(the reference runs these on CPU tensors) or on the GPU; the result comes back on the device
of the input.  There is no torch fallback: without ``libdiart_amd.so`` and a GPU they raise.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from . import _lib


def _gpu(device: Optional[torch.device] = None) -> torch.device:
    if device is not None and device.type == "cuda":
        return torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
    if not torch.cuda.is_available():
        raise _lib.DiartAmdError("diart_amd.functional needs an MI355X GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def overlapped_speech_penalty(segmentation: torch.Tensor, gamma: float = 3, beta: float = 10,
                              normalize: bool = False, speaker_major: bool = False) -> torch.Tensor:
    """segmentation (batch, frames, speakers) -> weights (paper Eq. 2), same shape —
    or (batch, speakers, frames) with ``speaker_major`` (the layout the pooling kernel reads).
    ``normalize`` adds the per-(batch, speaker) min-max of ``blocks/embedding.py:102-106``."""
    if segmentation.ndim != 3:
        raise ValueError("segmentation must be (batch, frames, speakers)")
    src = segmentation.device
    dev = _gpu(src)
    seg = segmentation.to(dev, torch.float32).contiguous()
    B, F, K = seg.shape
    out = torch.empty((B, K, F) if speaker_major else (B, F, K), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().dz_osp(_lib.context(dev.index), seg.data_ptr(), B, F, K, float(gamma),
                                  float(beta), int(bool(normalize)), int(bool(speaker_major)),
                                  out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dz_osp")
    return out if src == dev else out.to(src)


def normalize_embeddings(embeddings: torch.Tensor, norm: Union[float, torch.Tensor] = 1) -> torch.Tensor:
    """(batch, speakers, feat) or (speakers, feat) -> 3-D tensor with L2 norm ``norm``."""
    if embeddings.ndim == 2:
        embeddings = embeddings.unsqueeze(0)
    if isinstance(norm, torch.Tensor):
        b1, s1, _ = norm.shape
        b2, s2, _ = embeddings.shape
        assert b1 == b2 and s1 == s2
    src = embeddings.device
    dev = _gpu(src)
    out = embeddings.to(dev, torch.float32).contiguous().clone()
    B, K, D = out.shape
    _lib.check(_lib.load().dz_l2_normalize(_lib.context(dev.index), out.data_ptr(), B * K, D, 1.0,
                                           torch.cuda.current_stream(dev).cuda_stream), "dz_l2_normalize")
    if isinstance(norm, torch.Tensor):
        out = norm.to(dev) * out
    elif norm != 1:
        out = norm * out
    return out if src == dev else out.to(src)
