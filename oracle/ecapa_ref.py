"""PyTorch-CPU fp32 restatement of the config-3 embedding path (TEST INFRASTRUCTURE).

``pyannote.audio.pipelines.speaker_verification.PretrainedSpeakerEmbedding`` wrapping
``speechbrain/spkrec-ecapa-voxceleb`` — reached from the reference by the fallback at
``/root/reference/src/diart/models.py:59`` and called at ``models.py:262``; SURVEY.md Appendix
A.3.  Neither pyannote.audio nor speechbrain is in ``/root/reference`` or installable here, so
this is written from the published architecture (ECAPA-TDNN, C=1024, 192-d; Fbank(80), sentence
mean normalisation) — **parity unpinned**; module names are chosen so ``state_dict()`` keys equal
the speechbrain checkpoint's (``blocks.1.res2net_block.blocks.0.conv.conv.weight`` ...), which is
also the key set ``diart_amd.synth.synth_ecapa_state`` produces and ``diart_amd.weights`` consumes.

Quirks restated on purpose (they change the numbers): reflect "same" padding at the edge of the
PADDED batch tensor (so a row's embedding depends on the longest row of its batch), ``top_db``
clipping against the per-row maximum over valid AND padded frames, sentence mean over
``round(len * T)`` frames, masked SE mean / attentive statistics, rows with fewer than
``MIN_NUM_SAMPLES`` kept samples -> NaN, numpy output.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SAMPLE_RATE, N_FFT, HOP, N_MELS = 16000, 400, 160, 80
# PretrainedSpeakerEmbedding.min_num_samples bisects for the shortest input the network accepts:
# reflect padding of 4 frames (k=3, dilation 4) needs T = 1 + n // 160 >= 5 frames
MIN_NUM_SAMPLES = 640


# --------------------------------------------------------------------------- #
# features: speechbrain Fbank(n_mels=80) + InputNormalization("sentence", std_norm=False)
# --------------------------------------------------------------------------- #
def mel_filterbank(n_mels: int = N_MELS, n_fft: int = N_FFT, sample_rate: int = SAMPLE_RATE) -> torch.Tensor:
    """speechbrain Filterbank (triangular, f_min=0, f_max=sr/2): (n_fft//2+1, n_mels)."""
    to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
    mel = torch.linspace(to_mel(0.0), to_mel(sample_rate / 2), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    n_stft = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_stft)
    slope = (all_freqs.repeat(n_mels, 1) - f_central[:, None]) / band[:, None]
    fb = torch.max(torch.zeros(1), torch.min(slope + 1.0, -slope + 1.0))
    return fb.t().contiguous()          # (201, 80)


def fbank(wavs: torch.Tensor) -> torch.Tensor:
    """(N, L) -> (N, 1 + L // 160, 80) log-mel in dB with top_db = 80."""
    window = torch.hamming_window(N_FFT)
    spec = torch.stft(wavs, N_FFT, HOP, N_FFT, window, center=True, pad_mode="constant",
                      normalized=False, onesided=True, return_complex=True)
    power = (spec.real ** 2 + spec.imag ** 2).transpose(1, 2)            # (N, T, 201)
    mel = power @ mel_filterbank()
    x_db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    floor = x_db.amax(dim=(-2, -1)) - 80.0
    return torch.max(x_db, floor.view(-1, 1, 1))


def sentence_mean_norm(feats: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    out = feats.clone()
    for i in range(feats.shape[0]):
        n = int(torch.round(lens[i] * feats.shape[1]).int())
        out[i] = feats[i] - feats[i, :n].mean(dim=0)
    return out


# --------------------------------------------------------------------------- #
# ECAPA-TDNN (speechbrain.lobes.models.ECAPA_TDNN)
# --------------------------------------------------------------------------- #
class _Conv(nn.Module):
    """speechbrain Conv1d(padding="same", padding_mode="reflect") on (N, C, T)."""

    def __init__(self, cin, cout, k, dilation=1):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, dilation=dilation)
        self.pad = dilation * (k - 1) // 2

    def forward(self, x):
        if self.pad:
            x = F.pad(x, (self.pad, self.pad), mode="reflect")
        return self.conv(x)


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = nn.BatchNorm1d(c)

    def forward(self, x):
        return self.norm(x)


class TDNNBlock(nn.Module):
    def __init__(self, cin, cout, k, dilation=1):
        super().__init__()
        self.conv, self.norm = _Conv(cin, cout, k, dilation), _BN(cout)

    def forward(self, x):
        return self.norm(F.relu(self.conv(x)))


class Res2NetBlock(nn.Module):
    def __init__(self, c, scale=8, k=3, dilation=1):
        super().__init__()
        self.scale = scale
        self.blocks = nn.ModuleList([TDNNBlock(c // scale, c // scale, k, dilation) for _ in range(scale - 1)])

    def forward(self, x):
        y, prev = [], None
        for i, xi in enumerate(torch.chunk(x, self.scale, dim=1)):
            if i == 0:
                yi = xi
            elif i == 1:
                yi = self.blocks[i - 1](xi)
            else:
                yi = self.blocks[i - 1](xi + prev)
            y.append(yi)
            prev = yi
        return torch.cat(y, dim=1)


def length_to_mask(length: torch.Tensor, max_len: int) -> torch.Tensor:
    return (torch.arange(max_len)[None, :] < length[:, None]).float()


class SEBlock(nn.Module):
    def __init__(self, c, se):
        super().__init__()
        self.conv1, self.conv2 = _Conv(c, se, 1), _Conv(se, c, 1)

    def forward(self, x, lengths):
        L = x.shape[-1]
        mask = length_to_mask(lengths * L, L).unsqueeze(1)
        s = (x * mask).sum(dim=2, keepdim=True) / mask.sum(dim=2, keepdim=True)
        s = torch.sigmoid(self.conv2(F.relu(self.conv1(s))))
        return s * x


class SERes2NetBlock(nn.Module):
    def __init__(self, c, dilation, scale=8, se=128):
        super().__init__()
        self.tdnn1 = TDNNBlock(c, c, 1)
        self.res2net_block = Res2NetBlock(c, scale, 3, dilation)
        self.tdnn2 = TDNNBlock(c, c, 1)
        self.se_block = SEBlock(c, se)

    def forward(self, x, lengths):
        r = x
        x = self.tdnn2(self.res2net_block(self.tdnn1(x)))
        return self.se_block(x, lengths) + r


class AttentiveStatisticsPooling(nn.Module):
    def __init__(self, c, att=128):
        super().__init__()
        self.tdnn = TDNNBlock(c * 3, att, 1)
        self.conv = _Conv(att, c, 1)

    @staticmethod
    def _stats(x, m, eps=1e-12):
        mean = (m * x).sum(2)
        std = torch.sqrt((m * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))
        return mean, std

    def forward(self, x, lengths):
        L = x.shape[-1]
        mask = length_to_mask(lengths * L, L).unsqueeze(1)
        total = mask.sum(dim=2, keepdim=True)
        mean, std = self._stats(x, mask / total)
        attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
        attn = self.conv(torch.tanh(self.tdnn(attn)))
        attn = attn.masked_fill(mask == 0, float("-inf"))
        attn = F.softmax(attn, dim=2)
        mean, std = self._stats(x, attn)
        return torch.cat((mean, std), dim=1).unsqueeze(2)


class EcapaTdnnRef(nn.Module):
    """(N, T, 80) features, relative lengths (N,) -> (N, 192)."""

    def __init__(self, channels=1024, lin_neurons=192):
        super().__init__()
        c = channels
        self.blocks = nn.ModuleList([TDNNBlock(N_MELS, c, 5, 1), SERes2NetBlock(c, 2), SERes2NetBlock(c, 3),
                                     SERes2NetBlock(c, 4)])
        self.mfa = TDNNBlock(3 * c, 3 * c, 1)
        self.asp = AttentiveStatisticsPooling(3 * c)
        self.asp_bn = _BN(6 * c)
        self.fc = _Conv(6 * c, lin_neurons, 1)

    def forward(self, feats, lengths, return_intermediate: bool = False):
        x = feats.transpose(1, 2)
        xl = []
        for i, layer in enumerate(self.blocks):
            x = layer(x) if i == 0 else layer(x, lengths)
            xl.append(x)
        cat = torch.cat(xl[1:], dim=1)
        m = self.mfa(cat)
        pooled = self.asp(m, lengths)
        out = self.fc(self.asp_bn(pooled)).squeeze(2)
        if return_intermediate:
            return out, {"block0": xl[0], "cat": cat, "mfa": m, "pooled": pooled.squeeze(2)}
        return out


class PretrainedSpeakerEmbeddingRef:
    """``__call__(waveforms (N,1,S), masks (N,F) | None) -> ndarray (N,192)`` with NaN rows."""

    dimension = 192

    def __init__(self, state: Optional[dict] = None):
        self.model = EcapaTdnnRef().eval()
        if state is not None:
            self.model.load_state_dict(state)

    def to(self, device):
        return self

    def select(self, waveforms: torch.Tensor, masks: Optional[torch.Tensor]):
        """mask -> (padded kept samples (N, Lmax), kept counts (N,))."""
        N, _, S = waveforms.shape
        wav = waveforms[:, 0, :]
        if masks is None:
            return wav, torch.full((N,), S, dtype=torch.long)
        imasks = F.interpolate(masks.unsqueeze(1).float(), size=S, mode="nearest").squeeze(1) > 0.5
        kept = [w[m] for w, m in zip(wav, imasks)]
        return nn.utils.rnn.pad_sequence(kept, batch_first=True), imasks.sum(dim=1)

    def __call__(self, waveforms: torch.Tensor, masks: Optional[torch.Tensor] = None) -> np.ndarray:
        with torch.no_grad():
            signals, wav_lens = self.select(waveforms, masks)
            N, max_len = signals.shape
            if max_len < MIN_NUM_SAMPLES:
                return np.full((N, self.dimension), np.nan, dtype=np.float32)
            too_short = wav_lens < MIN_NUM_SAMPLES
            rel = wav_lens.float() / max_len
            rel[too_short] = 1.0
            feats = sentence_mean_norm(fbank(signals), rel)
            emb = self.model(feats, rel).numpy().copy()
            emb[too_short.numpy()] = np.nan
            return emb
