"""Synthetic 16 kHz mono streams and seeded network weights.

There is no network access for datasets or the gated pyannote checkpoints
(``/root/reference/README.md:101-109``), so measurement and parity use:

* ``synth_stream``: band-limited-noise "speakers" with Markov on/off turns
  (SURVEY.md §8d config 1/2 generator);
* ``synth_segmentation_state`` / ``synth_embedding_state``: random-init weights
  of exactly the published architectures, keyed like the pyannote checkpoints
  (``sincnet.conv1d.0.filterbank.low_hz_`` ...) so a real state dict can be
  substituted without touching any other code.

The scales below are not PyTorch's defaults: default init collapses the
segmentation output to a constant ~0.5, which would never exercise the
clustering thresholds.  Larger recurrent / classifier gains give activations
that sweep (0, 1) over time and differ between speakers.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

SAMPLE_RATE = 16000


# --------------------------------------------------------------------------- #
# audio
# --------------------------------------------------------------------------- #
def synth_stream(seed: int, seconds: float, num_speakers: int = 3,
                 sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """One mono float32 stream in [-1, 1], shape (samples,)."""
    from scipy import fft as sfft
    rng = np.random.default_rng(seed)
    n = int(round(seconds * sample_rate))
    nfft = sfft.next_fast_len(n, real=True)
    out = np.zeros(n, dtype=np.float64)
    t = np.arange(n) / sample_rate
    hop = sample_rate // 10  # state changes every 100 ms
    freqs = sfft.rfftfreq(nfft, 1.0 / sample_rate)
    for spk in range(num_speakers):
        # carrier: noise shaped by a speaker-specific comb of formant-like bands
        noise = rng.standard_normal(nfft, dtype=np.float32)
        spec = sfft.rfft(noise)
        env = np.zeros_like(freqs)
        f0 = 90.0 + 60.0 * spk + 20.0 * rng.random()
        for c in (f0 * 3, 500 + 230 * spk, 1500 + 310 * spk, 2600 + 170 * spk):
            env += np.exp(-0.5 * ((freqs - c) / (80.0 + 40.0 * spk)) ** 2)
        voiced = sfft.irfft(spec * env.astype(np.float32), nfft)[:n].astype(np.float64)
        voiced /= (np.abs(voiced).max() + 1e-9)
        voiced *= 0.5 * (1.0 + np.sin(2 * math.pi * f0 * t))  # glottal-ish AM
        # Markov on/off turns, mean turn ~2 s, mean pause ~3 s
        state, gate = rng.random() < 0.4, np.zeros(n)
        for h in range(0, n, hop):
            p_flip = 0.05 if state else 0.033
            if rng.random() < p_flip:
                state = not state
            gate[h:h + hop] = 1.0 if state else 0.0
        k = np.hanning(801)
        from scipy.signal import fftconvolve
        gate = fftconvolve(gate, k / k.sum(), mode="same")
        out += 0.35 * gate * voiced
    out += 0.003 * rng.standard_normal(n)
    return np.clip(out, -1.0, 1.0).astype(np.float32)


def synth_streams(num_streams: int, seconds: float, seed0: int = 0) -> np.ndarray:
    """(num_streams, samples); stream i uses seed seed0 + i.  Generated on host threads."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        return np.stack(list(ex.map(lambda i: synth_stream(seed0 + i, seconds), range(num_streams))))


def sliding_chunks(stream: np.ndarray, duration: float = 5.0, step: float = 0.5,
                   sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """All full windows of one stream, shape (num_chunks, samples) (a strided view)."""
    S, H = int(round(duration * sample_rate)), int(round(step * sample_rate))
    n = (stream.shape[-1] - S) // H + 1
    return np.lib.stride_tricks.sliding_window_view(stream, S)[::H][:n]


# --------------------------------------------------------------------------- #
# weights
# --------------------------------------------------------------------------- #
def _u(g, shape, bound):
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


def _sincnet_state(g: torch.Generator, prefix: str = "sincnet.") -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    to_mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
    to_hz = lambda mel: 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    mel = np.linspace(to_mel(30.0), to_mel(8000.0 - 100.0), 41, dtype="float32")
    hz = to_hz(mel).astype("float32")
    low = torch.from_numpy(hz[:-1].copy()).view(-1, 1)
    band = torch.from_numpy(np.diff(hz).astype("float32")).view(-1, 1)
    # perturb like a trained bank (a few negative values exercise the abs())
    low = low * (1.0 + 0.05 * torch.randn(low.shape, generator=g))
    band = band * (1.0 + 0.10 * torch.randn(band.shape, generator=g))
    low[3] = -low[3]
    band[7] = -band[7]
    sd[prefix + "wav_norm1d.weight"] = torch.tensor([1.0 + 0.2 * torch.randn((), generator=g).item()])
    sd[prefix + "wav_norm1d.bias"] = torch.tensor([0.05 * torch.randn((), generator=g).item()])
    sd[prefix + "conv1d.0.filterbank.low_hz_"] = low.float()
    sd[prefix + "conv1d.0.filterbank.band_hz_"] = band.float()
    sd[prefix + "conv1d.0.filterbank.window_"] = torch.from_numpy(np.hamming(251)[:125]).float()
    sd[prefix + "conv1d.0.filterbank.n_"] = (2 * math.pi * torch.arange(-125, 0.0).view(1, -1) / 16000.0)
    for i, (cin, cout) in ((1, (80, 60)), (2, (60, 60))):
        b = 1.5 / math.sqrt(cin * 5)
        sd[prefix + f"conv1d.{i}.weight"] = _u(g, (cout, cin, 5), b)
        sd[prefix + f"conv1d.{i}.bias"] = _u(g, (cout,), b)
    for i, c in enumerate((80, 60, 60)):
        sd[prefix + f"norm1d.{i}.weight"] = 1.0 + 0.3 * _u(g, (c,), 1.0)
        sd[prefix + f"norm1d.{i}.bias"] = 0.2 * _u(g, (c,), 1.0)
    return sd


def synth_segmentation_state(seed: int = 1234, num_speakers: int = 3,
                             powerset: bool = False) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = _sincnet_state(g)
    H = 128
    for layer in range(4):
        cin = 60 if layer == 0 else 2 * H
        for suf in ("", "_reverse"):
            bi = (2.0 if layer == 0 else 6.0) / math.sqrt(cin)
            bh = 1.6 / math.sqrt(H)
            sd[f"lstm.weight_ih_l{layer}{suf}"] = _u(g, (4 * H, cin), bi)
            sd[f"lstm.weight_hh_l{layer}{suf}"] = _u(g, (4 * H, H), bh)
            sd[f"lstm.bias_ih_l{layer}{suf}"] = _u(g, (4 * H,), 0.3)
            sd[f"lstm.bias_hh_l{layer}{suf}"] = _u(g, (4 * H,), 0.3)
    sd["linear.0.weight"] = _u(g, (128, 256), 5.0 / math.sqrt(256))
    sd["linear.0.bias"] = _u(g, (128,), 0.1)
    sd["linear.1.weight"] = _u(g, (128, 128), 4.0 / math.sqrt(128))
    sd["linear.1.bias"] = _u(g, (128,), 0.1)
    out = 7 if powerset else num_speakers
    sd["classifier.weight"] = _u(g, (out, 128), 6.0 / math.sqrt(128))
    sd["classifier.bias"] = _u(g, (out,), 0.5) - (0.0 if powerset else 0.6)
    return sd


def synth_embedding_state(seed: int = 4321, dimension: int = 512) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = _sincnet_state(g)
    tdnn = [(60, 512, 5), (512, 512, 3), (512, 512, 3), (512, 512, 1), (512, 1500, 1)]
    for i, (cin, cout, k) in enumerate(tdnn):
        b = 1.7 / math.sqrt(cin * k)
        sd[f"tdnns.{3 * i}.weight"] = _u(g, (cout, cin, k), b)
        sd[f"tdnns.{3 * i}.bias"] = _u(g, (cout,), b)
        sd[f"tdnns.{3 * i + 2}.weight"] = 1.0 + 0.3 * _u(g, (cout,), 1.0)
        sd[f"tdnns.{3 * i + 2}.bias"] = 0.1 * _u(g, (cout,), 1.0)
        sd[f"tdnns.{3 * i + 2}.running_mean"] = 0.1 * _u(g, (cout,), 1.0)
        sd[f"tdnns.{3 * i + 2}.running_var"] = 0.6 + 0.8 * torch.rand((cout,), generator=g)
        sd[f"tdnns.{3 * i + 2}.num_batches_tracked"] = torch.tensor(1000)
    sd["embedding.weight"] = _u(g, (dimension, 3000), 1.0 / math.sqrt(3000))
    sd["embedding.bias"] = _u(g, (dimension,), 1.0 / math.sqrt(3000))
    return sd


def _sincnet_spec(prefix: str = "sincnet."):
    f32 = torch.float32
    spec = [(prefix + "wav_norm1d.weight", (1,), f32), (prefix + "wav_norm1d.bias", (1,), f32),
            (prefix + "conv1d.0.filterbank.low_hz_", (40, 1), f32),
            (prefix + "conv1d.0.filterbank.band_hz_", (40, 1), f32),
            (prefix + "conv1d.0.filterbank.window_", (125,), f32),
            (prefix + "conv1d.0.filterbank.n_", (1, 125), f32)]
    for i, (cin, cout) in ((1, (80, 60)), (2, (60, 60))):
        spec += [(prefix + f"conv1d.{i}.weight", (cout, cin, 5), f32), (prefix + f"conv1d.{i}.bias", (cout,), f32)]
    for i, c in enumerate((80, 60, 60)):
        spec += [(prefix + f"norm1d.{i}.weight", (c,), f32), (prefix + f"norm1d.{i}.bias", (c,), f32)]
    return spec


def segmentation_spec(num_speakers: int = 3, powerset: bool = False):
    """[(key, shape, dtype)] of a pyannote/segmentation state dict, from the architecture alone
     — what every rank needs to know to receive the broadcast of the
    weights (``distributed.broadcast_state``) without loading or synthesising them itself."""
    f32, H = torch.float32, 128
    spec = _sincnet_spec()
    for layer in range(4):
        cin = 60 if layer == 0 else 2 * H
        for suf in ("", "_reverse"):
            spec += [(f"lstm.weight_ih_l{layer}{suf}", (4 * H, cin), f32), (f"lstm.weight_hh_l{layer}{suf}", (4 * H, H), f32),
                     (f"lstm.bias_ih_l{layer}{suf}", (4 * H,), f32), (f"lstm.bias_hh_l{layer}{suf}", (4 * H,), f32)]
    out = 7 if powerset else num_speakers
    spec += [("linear.0.weight", (128, 256), f32), ("linear.0.bias", (128,), f32),
             ("linear.1.weight", (128, 128), f32), ("linear.1.bias", (128,), f32),
             ("classifier.weight", (out, 128), f32), ("classifier.bias", (out,), f32)]
    return spec


def embedding_spec(dimension: int = 512):
    """[(key, shape, dtype)] of a pyannote/embedding (XVectorSincNet) state dict (Appendix A.2)."""
    f32 = torch.float32
    spec = _sincnet_spec()
    for i, (cin, cout, k) in enumerate([(60, 512, 5), (512, 512, 3), (512, 512, 3), (512, 512, 1), (512, 1500, 1)]):
        spec += [(f"tdnns.{3 * i}.weight", (cout, cin, k), f32), (f"tdnns.{3 * i}.bias", (cout,), f32),
                 (f"tdnns.{3 * i + 2}.weight", (cout,), f32), (f"tdnns.{3 * i + 2}.bias", (cout,), f32),
                 (f"tdnns.{3 * i + 2}.running_mean", (cout,), f32), (f"tdnns.{3 * i + 2}.running_var", (cout,), f32),
                 (f"tdnns.{3 * i + 2}.num_batches_tracked", (), torch.int64)]
    spec += [("embedding.weight", (dimension, 3000), f32), ("embedding.bias", (dimension,), f32)]
    return spec


def synth_ecapa_state(seed: int = 777, channels: int = 1024, lin_neurons: int = 192) -> Dict[str, torch.Tensor]:
    """Random-init weights of speechbrain's ECAPA-TDNN (spkrec-ecapa-voxceleb geometry), keyed like
    its checkpoint (``blocks.0.conv.conv.weight``, ``blocks.1.res2net_block.blocks.0.norm.norm.
    running_var``, ``asp.tdnn...``, ``fc.conv.weight``).  He-style bounds keep the ReLU stacks at
    unit scale so that every layer contributes to the output."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    c = channels

    def conv(prefix, cin, cout, k, gain=1.0):
        b = gain * math.sqrt(6.0 / (cin * k))
        sd[prefix + ".conv.weight"] = _u(g, (cout, cin, k), b)
        sd[prefix + ".conv.bias"] = _u(g, (cout,), 0.1)

    def bn(prefix, n):
        sd[prefix + ".norm.weight"] = 1.0 + 0.2 * _u(g, (n,), 1.0)
        sd[prefix + ".norm.bias"] = 0.1 * _u(g, (n,), 1.0)
        sd[prefix + ".norm.running_mean"] = 0.3 + 0.1 * _u(g, (n,), 1.0)
        sd[prefix + ".norm.running_var"] = 0.8 + 0.8 * torch.rand((n,), generator=g)
        sd[prefix + ".norm.num_batches_tracked"] = torch.tensor(1000)

    def tdnn(prefix, cin, cout, k, gain=0.55):
        conv(prefix + ".conv", cin, cout, k, gain)
        bn(prefix + ".norm", cout)

    tdnn("blocks.0", 80, c, 5, gain=0.08)            # input is log-mel in dB (tens of units)
    for i in (1, 2, 3):
        p = f"blocks.{i}"
        tdnn(p + ".tdnn1", c, c, 1)
        for j in range(7):
            tdnn(p + f".res2net_block.blocks.{j}", c // 8, c // 8, 3)
        tdnn(p + ".tdnn2", c, c, 1)
        conv(p + ".se_block.conv1", c, 128, 1)
        conv(p + ".se_block.conv2", 128, c, 1)
    tdnn("mfa", 3 * c, 3 * c, 1)
    tdnn("asp.tdnn", 9 * c, 128, 1, gain=0.3)
    conv("asp.conv", 128, 3 * c, 1, gain=1.5)
    bn("asp_bn", 6 * c)
    conv("fc", 6 * c, lin_neurons, 1)
    return sd
