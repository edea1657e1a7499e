"""Operator API of the hot path: ``SegmentationModel`` / ``EmbeddingModel``.

Mirrors the duck-typed boundary of the reference (``/root/reference/src/diart/models.py``:
``LazyModel`` :112-139, ``SegmentationModel`` :142-198, ``EmbeddingModel`` :201-265 and the
"Custom models" contract of ``/root/reference/README.md:186-209``): a model wraps a *loader*
(``Callable[[], Callable]``); the loaded object answers ``__call__``, ``.to(device)`` and is
only ``.eval()``-ed when it is an ``nn.Module``.  Here the loaded objects are
``HipSegmentation`` / ``HipEmbedding``: thin handles on ``libdiart_amd.so`` — every FLOP of
the forward pass runs in hand-written HIP kernels, torch only owns the memory.

Because the callables follow the reference contract they can also be handed to the
reference's own classes unchanged::

    from diart.models import SegmentationModel            # the reference
    from diart_amd.models import SegmentationLoader
    seg = SegmentationModel(SegmentationLoader("seg_state.pt"))

Loaders are picklable and hold no HIP state until ``__call__`` (``Parallelize`` pickles the
config that holds them, ``/root/reference/src/diart/inference.py:527-555``).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Callable, Dict, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .weights import PackedEcapa, PackedEmbedding, PackedSegmentation

StateSource = Union[str, Path, Dict[str, torch.Tensor]]


def default_precision(given: Optional[str] = None) -> str:
    """Arithmetic of the GEMM-shaped layers: the ``precision=`` argument of a model, "f16x3" when it does not say
    (f32 operands split into two f16 numbers = 22 mantissa bits, three f16 MFMAs per product, f32
    accumulation: measured against the f32 CPU restatement of the networks it is indistinguishable
    from "f32", the exact-f32 MFMA path, and 1.8x faster end to end; weights.PRECISIONS,
    DESIGN.md 4.2); ``DZ_ENGINE=precision=...`` overrides both (config.py)."""
    from .config import setting
    from .weights import PRECISIONS
    p = setting("precision", given, "f16x3")
    if p not in PRECISIONS:
        raise ValueError(f"precision={p!r}: expected one of {PRECISIONS}")
    return p


def _read_state(src: StateSource) -> Dict[str, torch.Tensor]:
    """A state dict, or a file holding one: ``.safetensors``, a plain ``torch.save`` of a state dict (speechbrain's
    ``embedding_model.ckpt``) or a PyTorch-Lightning checkpoint (``pyannote/segmentation``, ``pyannote/embedding``:
    what the reference loads through pyannote.audio, /root/reference/src/diart/models.py:50, :59).  Files are read
    by ``checkpoint.read_state``: tensors only, no foreign class is imported and no pickle code runs."""
    if isinstance(src, dict):
        return src
    from .checkpoint import read_state
    return read_state(src)


def _device_index(device: torch.device) -> int:
    if device.type != "cuda":
        raise _lib.DiartAmdError(
            f"diart_amd runs on MI355X only (requested device '{device}'); there is no CPU path")
    return device.index if device.index is not None else torch.cuda.current_device()


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _as_rows(waveform: torch.Tensor) -> torch.Tensor:
    """(B,1,S) or (B,S) float32 device tensor with 16-byte aligned rows; never copies a view
    that is already usable (so a rolling window is addressed in place)."""
    if waveform.ndim == 3:
        if waveform.shape[1] != 1:
            raise ValueError(f"expected mono audio (batch, 1, samples), got {tuple(waveform.shape)}")
        waveform = waveform[:, 0, :]
    if waveform.ndim != 2:
        raise ValueError(f"expected (batch, channels, samples), got {tuple(waveform.shape)}")
    if waveform.dtype != torch.float32:
        waveform = waveform.float()
    ok = (waveform.stride(1) == 1 and waveform.data_ptr() % 16 == 0
          and (waveform.shape[0] == 1 or waveform.stride(0) % 4 == 0))
    return waveform if ok else waveform.contiguous()


class _HipModule:
    """Common handle management: packed weights per device, one C handle per (S, max_batch)."""

    def __init__(self, state: Dict[str, torch.Tensor], max_batch: int):
        self._state = state
        self._max_batch = int(max_batch)
        self.device: Optional[torch.device] = None
        self._packed = None
        self._handles: Dict[int, tuple] = {}

    def to(self, device: Union[torch.device, str]):
        device = torch.device(device)
        idx = _device_index(device)
        device = torch.device("cuda", idx)
        if self.device != device:
            self._release()
            self.device = device
            self._packed = self._pack(device)
        return self

    def _need(self, num_samples: int, batch: int):
        if self.device is None:
            self.to(torch.device("cuda"))
        h = self._handles.get(num_samples)
        if h is None or h[1] < batch:
            if h is not None:
                self._destroy(h[0])
            cap = max(self._max_batch, batch)
            h = (self._create(num_samples, cap), cap)
            self._handles[num_samples] = h
        return h[0]

    def _release(self):
        for h, _ in self._handles.values():
            self._destroy(h)
        self._handles = {}
        self._packed = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # pickling ships the (CPU) state dict only
    def __getstate__(self):
        return {"_state": {k: v.cpu() for k, v in self._state.items()},
                "_max_batch": self._max_batch, "extra": self._extra_state()}

    def __setstate__(self, st):
        self.__init__(st["_state"], st["_max_batch"], **st["extra"])


class HipSegmentation(_HipModule):
    """pyannote/segmentation forward: ``waveform (B,1,S) -> (B,F,K)`` activations in [0,1]
    (hard {0,1} multilabel when ``powerset``) — the callable behind
    ``SegmentationModel.__call__`` (reference models.py:188-198)."""

    def __init__(self, state: Dict[str, torch.Tensor], max_batch: int = 64, powerset: bool = False,
                 precision: Optional[str] = None, recurrence: Optional[str] = None):
        """``recurrence``: the LSTM recurrence kernel of this model's own calls — "valu" (default: one chain per CU on
        the f32 vector units, the shortest layer) or a matrix-core variant "0" | "3" | "4" (16 chains per workgroup,
        default precision only).  A throughput engine (``StreamBatch``) chooses its own (``struct_throughput``)."""
        super().__init__(state, max_batch)
        from .config import setting
        self.powerset = bool(powerset)
        self.precision = default_precision(precision)
        self.recurrence = str(setting("recurrence", recurrence, "valu"))
        self.num_speakers: Optional[int] = None

    def _extra_state(self):
        return {"powerset": self.powerset, "precision": self.precision, "recurrence": self.recurrence}

    def _pack(self, device):
        p = PackedSegmentation(self._state, device, powerset=self.powerset, precision=self.precision,
                               recurrence=self.recurrence)
        self.num_speakers = p.num_speakers
        return p

    def _create(self, num_samples, cap, throughput: bool = False, recurrence: Optional[str] = None):
        """``throughput``: the handle of an engine that keeps several steps in flight (``StreamBatch``): the matrix-core
        recurrence (``PackedSegmentation.struct_throughput``) instead of the low-latency one; ``recurrence``: that
        kernel, whatever the model's own (``PackedSegmentation.struct_for``)."""
        h = _lib.vp()
        lib = _lib.load()
        w = (self._packed.struct_for(recurrence) if recurrence is not None else
             self._packed.struct_throughput if throughput else self._packed.struct)
        _lib.check(lib.dz_seg_create(_lib.context(self.device.index), C.byref(w),
                                     cap, num_samples, C.byref(h)), "dz_seg_create")
        return h

    def throughput_recurrence(self) -> str:
        """What a throughput handle of this model runs its recurrence on: "valu" | "0" | "3" | "4" (bench.py names the kernel)."""
        p = self._packed
        return str(int(p.struct_throughput.lstm_variant)) if p.struct_throughput.whh_split[0] else "valu"

    def _destroy(self, h):
        _lib.load().dz_seg_destroy(h)

    def num_frames(self, num_samples: int) -> int:
        return int(_lib.load().dz_seg_frames_for(int(num_samples)))

    def __call__(self, waveform: torch.Tensor) -> torch.Tensor:
        if self.device is None:
            self.to(waveform.device)
        if waveform.device != self.device:
            waveform = waveform.to(self.device)
        rows = _as_rows(waveform)
        B, S = rows.shape
        if B < 1:
            raise ValueError("empty batch")
        handle = self._need(S, B)
        out = torch.empty((B, self.num_frames(S), self.num_speakers), dtype=torch.float32,
                          device=self.device)
        _lib.check(_lib.load().dz_seg_forward(handle, rows.data_ptr(), rows.stride(0) if B > 1 else S,
                                              B, out.data_ptr(), _stream_ptr(self.device)),
                   "dz_seg_forward")
        return out


class HipEmbedding(_HipModule):
    """pyannote/embedding forward: ``(waveform (N,1,S), weights (N,F) | None) -> (N,512)`` — the
    callable behind ``EmbeddingModel.__call__`` (reference models.py:248-265)."""

    dimension = 512

    def __init__(self, state: Dict[str, torch.Tensor], max_batch: int = 64, precision: Optional[str] = None,
                 weight_interp: Optional[str] = None):
        """``weight_interp``: StatsPool's resampling of the pooling weights to the feature frames — "linear"
        (pyannote.audio 2.x .. 3.0: ``F.interpolate(mode="linear")``; the default, setup.cfg pins ``>=2.1.1``) or
        "nearest" (pyannote.audio >= 3.1).  ``EmbeddingLoader`` sets it from the version a checkpoint records."""
        super().__init__(state, max_batch)
        self.precision = default_precision(precision)
        self.weight_interp = weight_interp or "linear"

    def _extra_state(self):
        return {"precision": self.precision, "weight_interp": self.weight_interp}

    def _pack(self, device):
        return PackedEmbedding(self._state, device, precision=self.precision, weight_interp=self.weight_interp)

    def _create(self, num_samples, cap):
        h = _lib.vp()
        _lib.check(_lib.load().dz_emb_create(_lib.context(self.device.index),
                                             C.byref(self._packed.struct), cap, num_samples,
                                             C.byref(h)), "dz_emb_create")
        return h

    def _destroy(self, h):
        _lib.load().dz_emb_destroy(h)

    def __call__(self, waveform: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.device is None:
            self.to(waveform.device)
        rows = _as_rows(waveform.to(self.device))
        N, S = rows.shape
        wptr, fw = None, 0
        if weights is not None:
            weights = weights.to(self.device, torch.float32).contiguous()
            if weights.ndim != 2 or weights.shape[0] != N:
                raise ValueError(f"weights must be (batch, frames), got {tuple(weights.shape)}")
            wptr, fw = weights.data_ptr(), weights.shape[1]
        handle = self._need(S, N)
        out = torch.empty((N, self.dimension), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().dz_emb_forward(handle, rows.data_ptr(), rows.stride(0) if N > 1 else S,
                                              wptr, N, fw, out.data_ptr(), _stream_ptr(self.device)),
                   "dz_emb_forward")
        return out

    def forward_multi(self, waveform: torch.Tensor, weights: torch.Tensor,
                      normalize: bool = False) -> torch.Tensor:
        """``waveform (B,1,S)``, ``weights (B,K,F)`` speaker-major -> ``(B,K,512)``.

        Same result as the reference's ``(B*K)``-row call (blocks/embedding.py:56-65) with
        the speaker-independent frame features computed once per chunk instead of K times.
        """
        if self.device is None:
            self.to(waveform.device)
        rows = _as_rows(waveform.to(self.device))
        B, S = rows.shape
        weights = weights.to(self.device, torch.float32).contiguous()
        if weights.ndim != 3 or weights.shape[0] != B:
            raise ValueError(f"weights must be (batch, speakers, frames), got {tuple(weights.shape)}")
        K, fw = weights.shape[1], weights.shape[2]
        handle = self._need(S, B)
        out = torch.empty((B, K, self.dimension), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().dz_emb_forward_multi(handle, rows.data_ptr(),
                                                    rows.stride(0) if B > 1 else S,
                                                    weights.data_ptr(), B, K, fw,
                                                    1 if normalize else 0, out.data_ptr(),
                                                    _stream_ptr(self.device)), "dz_emb_forward_multi")
        return out


class HipEcapaEmbedding(_HipModule):
    """speechbrain ECAPA-TDNN behind pyannote's ``PretrainedSpeakerEmbedding`` contract
    (BASELINE.json config 3): ``(waveform (N,1,S), masks (N,F) | None) -> (N,192)``; a row whose
    mask keeps fewer than 640 samples is NaN.  The reference reaches this model through the
    fallback at models.py:59 and calls it at :262; it returns numpy there, a device tensor here
    (``EmbeddingModel`` accepts both, models.py:263-264)."""

    dimension = 192

    def __init__(self, state: Dict[str, torch.Tensor], max_batch: int = 192, precision: Optional[str] = None):
        super().__init__(state, max_batch)
        self.precision = default_precision(precision)

    def _extra_state(self):
        return {"precision": self.precision}

    def _pack(self, device):
        return PackedEcapa(self._state, device, precision=self.precision)

    def _create(self, num_samples, cap):
        h = _lib.vp()
        _lib.check(_lib.load().dz_ecapa_create(_lib.context(self.device.index),
                                               C.byref(self._packed.struct), cap, num_samples,
                                               C.byref(h)), "dz_ecapa_create")
        return h

    def _destroy(self, h):
        _lib.load().dz_ecapa_destroy(h)

    def __call__(self, waveform: torch.Tensor, masks: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.device is None:
            self.to(waveform.device)
        rows = _as_rows(waveform.to(self.device))
        N, S = rows.shape
        mptr, fw = None, 0
        if masks is not None:
            masks = masks.to(self.device, torch.float32).contiguous()
            if masks.ndim != 2 or masks.shape[0] != N:
                raise ValueError(f"masks must be (batch, frames), got {tuple(masks.shape)}")
            mptr, fw = masks.data_ptr(), masks.shape[1]
        handle = self._need(S, N)
        out = torch.empty((N, self.dimension), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().dz_ecapa_forward(handle, rows.data_ptr(), rows.stride(0) if N > 1 else S,
                                                mptr, N, fw, out.data_ptr(), _stream_ptr(self.device)),
                   "dz_ecapa_forward")
        return out

    def last_frames(self, num_samples: int) -> int:
        """Frames of the batch geometry of the last forward (= those of its longest kept row, what every row is
        padded to: ``dz_ecapa_peek``); no copy, no synchronisation.  ``bench.py --config 3`` prices its kernels with it."""
        ptr, cnt, frames = _lib.vp(), C.c_longlong(), C.c_int()
        _lib.check(_lib.load().dz_ecapa_peek(self._handles[num_samples][0], 5, C.byref(ptr),
                                             C.byref(cnt), C.byref(frames)), "dz_ecapa_peek")
        return frames.value

    def peek(self, num_samples: int, which: int) -> torch.Tensor:
        """Intermediate of the last forward (parity tests): see ``dz_ecapa_peek``."""
        ptr, cnt, frames = _lib.vp(), C.c_longlong(), C.c_int()
        _lib.check(_lib.load().dz_ecapa_peek(self._handles[num_samples][0], which, C.byref(ptr),
                                             C.byref(cnt), C.byref(frames)), "dz_ecapa_peek")
        dtype = torch.int32 if which == 5 else torch.float32
        out = torch.empty(cnt.value, dtype=dtype, device=self.device)
        torch.cuda.synchronize(self.device)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ptr, ctypes.c_size_t(cnt.value * 4), 3)
        if rc != 0:
            raise _lib.DiartAmdError(f"hipMemcpy failed ({rc})")
        return out, frames.value


# --------------------------------------------------------------------------- #
# loaders (picklable, no HIP state)
# --------------------------------------------------------------------------- #
class SegmentationLoader:
    def __init__(self, state: StateSource, max_batch: int = 64, powerset: bool = False,
                 precision: Optional[str] = None):
        self.state, self.max_batch, self.powerset, self.precision = state, max_batch, powerset, precision

    def __call__(self) -> HipSegmentation:
        return HipSegmentation(_read_state(self.state), self.max_batch, self.powerset, self.precision)


class EmbeddingLoader:
    """``arch``: "xvector" (pyannote/embedding) or "ecapa" (speechbrain/spkrec-ecapa-voxceleb);
    None = decide from the checkpoint keys."""

    def __init__(self, state: StateSource, max_batch: int = 64, arch: Optional[str] = None,
                 precision: Optional[str] = None, weight_interp: Optional[str] = None):
        """``weight_interp`` (x-vector only): "linear" | "nearest" | None = from the ``pyannote.audio`` version the
        checkpoint file records (>= 3.1: "nearest"; older, absent, or a plain state dict: "linear")."""
        self.state, self.max_batch, self.arch, self.precision = state, max_batch, arch, precision
        self.weight_interp = weight_interp

    def __call__(self):
        sd = _read_state(self.state)
        arch = self.arch or ("ecapa" if any(k.startswith("asp.") for k in sd) else "xvector")
        if arch == "ecapa":
            return HipEcapaEmbedding(sd, self.max_batch, self.precision)
        interp = self.weight_interp
        if interp is None and not isinstance(self.state, dict):
            from .checkpoint import pyannote_version
            v = pyannote_version(self.state)
            interp = "nearest" if v is not None and v >= (3, 1) else "linear"
        return HipEmbedding(sd, self.max_batch, self.precision, interp)


# --------------------------------------------------------------------------- #
# the reference's wrappers (same names, same semantics)
# --------------------------------------------------------------------------- #
class LazyModel:
    """Defers loading until first use; ``.to`` loads then moves; ``__call__`` forwards."""

    def __init__(self, loader: Callable[[], Callable]):
        self.get_model = loader
        self.model: Optional[Callable] = None

    def is_in_memory(self) -> bool:
        return self.model is not None

    def load(self):
        if self.model is None:
            self.model = self.get_model()

    def to(self, device: torch.device) -> "LazyModel":
        self.load()
        self.model = self.model.to(device)
        return self

    def eval(self) -> "LazyModel":
        self.load()
        if isinstance(self.model, nn.Module):
            self.model.eval()
        return self

    def __call__(self, *args, **kwargs):
        self.load()
        return self.model(*args, **kwargs)


def _no_onnx(*_a, **_k):
    raise NotImplementedError(
        "diart_amd replaces the torch/ONNX back-ends of the hot path with HIP kernels; "
        "ONNX models are not part of this path (reference models.py:62-109)")


class SegmentationModel(LazyModel):
    """``waveform (batch, channels, samples) -> (batch, frames, speakers)``."""

    from_onnx = staticmethod(_no_onnx)

    @staticmethod
    def from_state(state: StateSource, max_batch: int = 64, powerset: bool = False,
                   precision: Optional[str] = None) -> "SegmentationModel":
        return SegmentationModel(SegmentationLoader(state, max_batch, powerset, precision))

    @staticmethod
    def from_pyannote(model, use_hf_token=True) -> "SegmentationModel":
        """``model``: path to a pyannote checkpoint / state-dict file (the hub is unreachable
        from an air-gapped MI355X box, so hub names are rejected with a clear message)."""
        if isinstance(model, (str, Path)) and Path(model).exists():
            return SegmentationModel.from_state(model, powerset="3.0" in str(model))
        raise FileNotFoundError(
            f"'{model}': pass a local pyannote checkpoint / state-dict file "
            "(gated HuggingFace downloads are not available here)")

    @staticmethod
    def from_pretrained(model, use_hf_token=True) -> "SegmentationModel":
        if isinstance(model, (str, Path)) and Path(model).name.endswith(".onnx"):
            return SegmentationModel.from_onnx(model)
        if isinstance(model, dict):
            return SegmentationModel.from_state(model)
        return SegmentationModel.from_pyannote(model, use_hf_token)

    def __call__(self, waveform: torch.Tensor) -> torch.Tensor:
        return super().__call__(waveform)


class EmbeddingModel(LazyModel):
    """``(waveform (batch, channels, samples), weights (batch, frames) | None) ->
    (batch, embedding_dim)``; numpy results are converted like the reference does."""

    from_onnx = staticmethod(_no_onnx)

    @staticmethod
    def from_state(state: StateSource, max_batch: int = 64, arch: Optional[str] = None,
                   precision: Optional[str] = None, weight_interp: Optional[str] = None) -> "EmbeddingModel":
        return EmbeddingModel(EmbeddingLoader(state, max_batch, arch, precision, weight_interp))

    @staticmethod
    def from_pyannote(model, use_hf_token=True) -> "EmbeddingModel":
        if isinstance(model, (str, Path)) and Path(model).exists():
            return EmbeddingModel.from_state(model)
        raise FileNotFoundError(
            f"'{model}': pass a local pyannote checkpoint / state-dict file "
            "(gated HuggingFace downloads are not available here)")

    @staticmethod
    def from_pretrained(model, use_hf_token=True) -> "EmbeddingModel":
        if isinstance(model, (str, Path)) and Path(model).name.endswith(".onnx"):
            return EmbeddingModel.from_onnx(model)
        if isinstance(model, dict):
            return EmbeddingModel.from_state(model)
        return EmbeddingModel.from_pyannote(model, use_hf_token)

    def __call__(self, waveform: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = super().__call__(waveform, weights)
        if isinstance(out, np.ndarray):
            out = torch.from_numpy(out)
        return out
