"""PyTorch-CPU fp32 restatement of the two networks on diart's hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) — "parity unpinned" for the
network arithmetic: the code these modules restate lives in third-party
``pyannote.audio`` (>=2.1.1, ``/root/reference/setup.cfg:35``) and
``asteroid-filterbanks`` which are absent here.  Call sites in the reference:

* segmentation  ``/root/reference/src/diart/models.py:133`` via ``:188-198``
  (``PyanNet.forward``: SincNet -> 4xBiLSTM(128) -> 2xLinear(128) -> Linear(K)
  -> sigmoid)
* embedding     ``/root/reference/src/diart/models.py:262`` via ``:248-265``
  (``XVectorSincNet.forward(waveforms, weights)``: SincNet -> 5 TDNN ->
  weighted StatsPool -> Linear(3000, 512))
* powerset      ``/root/reference/src/diart/models.py:29-39``
  (``Powerset.to_multilabel``: hard argmax -> multilabel)

Module / parameter names are chosen so ``state_dict()`` keys equal the keys of
the pyannote checkpoints (``sincnet.conv1d.0.filterbank.low_hz_`` ...), which is
also the key set ``diart_amd.weights`` produces and consumes.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# SincNet front-end (pyannote.audio.models.blocks.sincnet.SincNet, stride=10)
# --------------------------------------------------------------------------- #
class ParamSincFBRef(nn.Module):
    """asteroid_filterbanks.ParamSincFB(n_filters=80, kernel_size=251).

    40 learnable (low_hz_, band_hz_) pairs -> 40 cos (symmetric) + 40 sin
    (antisymmetric) band-pass FIR filters of 251 taps.
    """

    def __init__(self, n_filters: int = 80, kernel_size: int = 251,
                 sample_rate: float = 16000.0, min_low_hz: float = 50.0,
                 min_band_hz: float = 50.0):
        super().__init__()
        assert kernel_size % 2 == 1 and n_filters % 2 == 0
        self.n_filters, self.kernel_size = n_filters, kernel_size
        self.sample_rate = sample_rate
        self.min_low_hz, self.min_band_hz = min_low_hz, min_band_hz
        self.half_kernel = kernel_size // 2
        cutoff = n_filters // 2
        # mel-spaced initialisation, as the library does
        to_mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
        to_hz = lambda mel: 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
        low_hz, high_hz = 30.0, sample_rate / 2 - (min_low_hz + min_band_hz)
        mel = np.linspace(to_mel(low_hz), to_mel(high_hz), cutoff + 1, dtype="float32")
        hz = to_hz(mel)
        self.low_hz_ = nn.Parameter(torch.from_numpy(hz[:-1].astype("float32")).view(-1, 1))
        self.band_hz_ = nn.Parameter(torch.from_numpy(np.diff(hz).astype("float32")).view(-1, 1))
        window_ = np.hamming(kernel_size)[: self.half_kernel]
        n_ = 2 * math.pi * (torch.arange(-self.half_kernel, 0.0).view(1, -1) / sample_rate)
        self.register_buffer("window_", torch.from_numpy(window_).float())
        self.register_buffer("n_", n_)

    def _make(self, low, high, kind):
        band = (high - low)[:, 0]
        ft_low = torch.matmul(low, self.n_)
        ft_high = torch.matmul(high, self.n_)
        if kind == "cos":
            left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (self.n_ / 2)) * self.window_
            center = 2 * band.view(-1, 1)
            right = torch.flip(left, dims=[1])
        else:
            left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (self.n_ / 2)) * self.window_
            center = torch.zeros_like(band.view(-1, 1))
            right = -torch.flip(left, dims=[1])
        bp = torch.cat([left, center, right], dim=1)
        return bp / (2 * band[:, None])

    def filters(self) -> torch.Tensor:
        low = self.min_low_hz + torch.abs(self.low_hz_)
        high = torch.clamp(low + self.min_band_hz + torch.abs(self.band_hz_),
                           self.min_low_hz, self.sample_rate / 2)
        return torch.cat([self._make(low, high, "cos"), self._make(low, high, "sin")],
                         dim=0).view(self.n_filters, 1, self.kernel_size)


class _SincEncoder(nn.Module):
    """asteroid Encoder(ParamSincFB) == conv1d with the generated filters."""

    def __init__(self, stride: int):
        super().__init__()
        self.filterbank = ParamSincFBRef()
        self.stride = stride

    def forward(self, x):
        return F.conv1d(x, self.filterbank.filters(), stride=self.stride)


class SincNetRef(nn.Module):
    def __init__(self, stride: int = 10):
        super().__init__()
        self.wav_norm1d = nn.InstanceNorm1d(1, affine=True)
        self.conv1d = nn.ModuleList([_SincEncoder(stride), nn.Conv1d(80, 60, 5), nn.Conv1d(60, 60, 5)])
        self.pool1d = nn.ModuleList([nn.MaxPool1d(3, stride=3) for _ in range(3)])
        self.norm1d = nn.ModuleList([nn.InstanceNorm1d(80, affine=True),
                                     nn.InstanceNorm1d(60, affine=True),
                                     nn.InstanceNorm1d(60, affine=True)])

    def forward(self, waveforms: torch.Tensor) -> torch.Tensor:
        out = self.wav_norm1d(waveforms)
        for c, (conv, pool, norm) in enumerate(zip(self.conv1d, self.pool1d, self.norm1d)):
            out = conv(out)
            if c == 0:
                out = torch.abs(out)
            out = F.leaky_relu(norm(pool(out)))
        return out  # (B, 60, 293) for 80000 samples


# --------------------------------------------------------------------------- #
# pyannote/segmentation  = PyanNet
# --------------------------------------------------------------------------- #
class PyanNetRef(nn.Module):
    """(B,1,S) -> (B,F,K) multilabel activations in [0,1] (or log-probs if powerset)."""

    def __init__(self, num_speakers: int = 3, powerset: bool = False):
        super().__init__()
        self.sincnet = SincNetRef(stride=10)
        self.lstm = nn.LSTM(60, 128, num_layers=4, bidirectional=True, batch_first=True)
        self.linear = nn.ModuleList([nn.Linear(256, 128), nn.Linear(128, 128)])
        self.powerset = powerset
        self.num_speakers = num_speakers
        out = 7 if powerset else num_speakers
        self.classifier = nn.Linear(128, out)

    def forward(self, waveforms: torch.Tensor) -> torch.Tensor:
        x = self.sincnet(waveforms)                 # (B,60,F)
        x, _ = self.lstm(x.transpose(1, 2))         # (B,F,256)
        for lin in self.linear:
            x = F.leaky_relu(lin(x))
        x = self.classifier(x)
        if self.powerset:
            return F.log_softmax(x, dim=-1)
        return torch.sigmoid(x)


def powerset_mapping(num_classes: int = 3, max_set_size: int = 2) -> torch.Tensor:
    """pyannote Powerset.mapping: rows = subsets ordered by size then lexicographic."""
    import itertools
    rows = []
    for size in range(max_set_size + 1):
        for subset in itertools.combinations(range(num_classes), size):
            row = torch.zeros(num_classes)
            row[list(subset)] = 1.0
            rows.append(row)
    return torch.stack(rows)  # (7,3) for (3,2)


def powerset_to_multilabel(logp: torch.Tensor, num_classes: int = 3, max_set_size: int = 2):
    """Powerset.to_multilabel (hard): one_hot(argmax) @ mapping — models.py:38-39."""
    mapping = powerset_mapping(num_classes, max_set_size).to(logp)
    hard = F.one_hot(torch.argmax(logp, dim=-1), mapping.shape[0]).to(logp.dtype)
    return hard @ mapping


# --------------------------------------------------------------------------- #
# pyannote/embedding = XVectorSincNet
# --------------------------------------------------------------------------- #
def stats_pool_ref(seq: torch.Tensor, weights: Optional[torch.Tensor],
                   interp_mode: str = "linear", eps_den: float = 0.0) -> torch.Tensor:
    """pyannote StatsPool: seq (N,C,T), weights (N,Fw) or None -> (N,2C). Paper Eq. 1."""
    if weights is None:
        return torch.cat([seq.mean(dim=2), seq.std(dim=2, unbiased=True)], dim=1)
    w = weights.unsqueeze(1)
    if w.shape[2] != seq.shape[2]:
        if interp_mode == "linear":
            w = F.interpolate(w, size=seq.shape[2], mode="linear", align_corners=False)
        else:
            w = F.interpolate(w, size=seq.shape[2], mode="nearest")
    v1 = w.sum(dim=2)
    mean = torch.sum(seq * w, dim=2) / v1
    dx2 = torch.square(seq - mean.unsqueeze(2))
    v2 = torch.square(w).sum(dim=2)
    var = torch.sum(dx2 * w, dim=2) / (v1 - v2 / v1 + eps_den)
    return torch.cat([mean, torch.sqrt(var)], dim=1)


class XVectorSincNetRef(nn.Module):
    TDNN = [(60, 512, 5, 1), (512, 512, 3, 2), (512, 512, 3, 3), (512, 512, 1, 1), (512, 1500, 1, 1)]

    def __init__(self, dimension: int = 512):
        super().__init__()
        self.sincnet = SincNetRef(stride=10)
        mods = []
        for cin, cout, k, d in self.TDNN:
            mods += [nn.Conv1d(cin, cout, k, dilation=d), nn.LeakyReLU(), nn.BatchNorm1d(cout)]
        self.tdnns = nn.ModuleList(mods)
        self.embedding = nn.Linear(3000, dimension)

    def frames(self, waveforms: torch.Tensor) -> torch.Tensor:
        x = self.sincnet(waveforms)
        for m in self.tdnns:
            x = m(x)
        return x  # (N,1500,279)

    def forward(self, waveforms: torch.Tensor, weights: Optional[torch.Tensor] = None, interp_mode: str = "linear"):
        return self.embedding(stats_pool_ref(self.frames(waveforms), weights, interp_mode=interp_mode))

    def forward_multi(self, waveforms: torch.Tensor, weights: torch.Tensor, interp_mode: str = "linear") -> torch.Tensor:
        """De-duplicated equivalent of the reference's (B*K)-row call.

        waveforms (B,1,S); weights (B,F,K) -> (B,K,D).  Mathematically identical
        to repeating each waveform K times (``blocks/embedding.py:57-59``) because
        only the pooling depends on the speaker.
        """
        fr = self.frames(waveforms)                          # (B,1500,279)
        B, Fw, K = weights.shape
        fr = fr.unsqueeze(1).expand(B, K, *fr.shape[1:]).reshape(B * K, *fr.shape[1:])
        w = weights.permute(0, 2, 1).reshape(B * K, Fw)
        return self.embedding(stats_pool_ref(fr, w, interp_mode=interp_mode)).view(B, K, -1)


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
