"""State dict (pyannote checkpoint key names) -> packed fp32 device tensors for the kernels.

PyTorch-ROCm only *holds* the weights (north-star: "PyTorch-ROCm holds only the weight
tensors"); all arithmetic on them happens in ``libdiart_amd.so``.  Packing happens once,
on the CPU, at load time:

* sinc FIR bank generated from ``low_hz_ / band_hz_`` (asteroid ``ParamSincFB.filters``,
  SURVEY.md Appendix A.1) and stored k-major ``[252][80]`` (tap 251 is a zero pad);
* conv / TDNN weights reordered ``[co][tap][ci]`` so an im2col row of channels-last
  activations is contiguous, channel counts padded 60 -> 64 and 1500 -> 1536 with zeros;
* ``BatchNorm1d`` (eval) folded to a per-channel scale / shift applied after LeakyReLU;
* LSTM: both directions' ``W_ih`` stacked to one ``[1024][K]`` GEMM operand, ``b_ih+b_hh``
  pre-summed, ``W_hh`` as ``[2][512][128]``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import _lib

BN_EPS = 1e-5
# arithmetic of the GEMM-shaped layers: "f32" = exact-f32 MFMA (v_mfma_f32_16x16x4_f32);
# "f16x3" = both operands split into (hi, lo) f16 pairs (22 mantissa bits), 3 f16 MFMAs per product, f32
# accumulation (k_gemm_split.hip; ~2^-16 relative error per product)
PRECISIONS = ("f32", "f16x3")


def sinc_filters(low_hz_: torch.Tensor, band_hz_: torch.Tensor, window_: torch.Tensor,
                 n_: torch.Tensor, sample_rate: float = 16000.0, min_low_hz: float = 50.0,
                 min_band_hz: float = 50.0) -> torch.Tensor:
    """[80][251] band-pass filters: 40 cos (even) then 40 sin (odd)."""
    low = min_low_hz + torch.abs(low_hz_.float())
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_.float()), min_low_hz, sample_rate / 2)
    band = (high - low)[:, 0]
    n_ = n_.float().view(1, -1)
    window_ = window_.float()
    ft_low, ft_high = torch.matmul(low, n_), torch.matmul(high, n_)
    out = []
    for kind in ("cos", "sin"):
        if kind == "cos":
            left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (n_ / 2)) * window_
            center = 2 * band.view(-1, 1)
            right = torch.flip(left, dims=[1])
        else:
            left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (n_ / 2)) * window_
            center = torch.zeros_like(band.view(-1, 1))
            right = -torch.flip(left, dims=[1])
        out.append(torch.cat([left, center, right], dim=1) / (2 * band[:, None]))
    return torch.cat(out, dim=0)


def fold_sinc_filters(filt: torch.Tensor) -> torch.Tensor:
    """[80][251] symmetric bank -> the folded k-major image ``[128][96]`` ``sinc_conv0`` consumes.

    Row ``j`` is the tap at distance ``j`` from the centre (tap 125); columns 0..39 the cos (even)
    filters, 48..87 the sin (odd) filters, the rest zero.  ``conv = sum_j cos_j (x[c+j] + x[c-j])
    + sin_j (x[c+j] - x[c-j])``: the centre tap of the even filters is halved (exact), that of
    the odd filters is zero.  The symmetry is what ``ParamSincFB.filters`` builds (``right =
    flip(left)`` / ``-flip(left)``); a bank that does not have it is refused."""
    assert filt.shape == (80, 251), filt.shape
    f = filt.detach().float().cpu()
    left, right = torch.flip(f[:, :125], dims=[1]), f[:, 126:]
    scale = float(f.abs().max()) + 1e-30
    if float((right[:40] - left[:40]).abs().max()) > 1e-6 * scale or \
            float((right[40:] + left[40:]).abs().max()) > 1e-6 * scale or \
            float(f[40:, 125].abs().max()) > 1e-6 * scale:
        raise ValueError("sinc filter bank is not (anti)symmetric around its centre tap")
    out = torch.zeros(128, 96, dtype=torch.float32)
    out[0, :40] = 0.5 * f[:40, 125]
    out[1:126, :40] = right[:40].t()
    out[1:126, 48:88] = right[40:].t()
    return out


def split_f16(w: torch.Tensor, name: Optional[str] = None) -> torch.Tensor:
    """f32 matrix ``[N][K]`` -> int16 ``[2][N][K]`` of IEEE f16 bit patterns: plane 0 ``hi = f16(w)``,
    plane 1 ``lo = f16((w - hi) * 2^11)`` — the two-term split ``k_gemm_split.hip`` multiplies with
    (``w = hi + lo * 2^-11`` to 22 mantissa bits; the scale keeps ``lo`` a normal f16)."""
    w = w.detach().float().cpu().contiguous()
    if w.numel() and not bool(torch.isfinite(w).all()):
        # (max() of a tensor with NaN is NaN and every comparison below would pass: the kernels' clamps would then turn
        # the NaN planes into finite garbage where an f32 model gives NaN outputs)
        raise ValueError(f"split_f16({name or 'matrix'}): {int((~torch.isfinite(w)).sum())} non-finite weights — a damaged "
                         "checkpoint; the \"f16x3\" arithmetic refuses it (precision=\"f32\" computes NaN outputs like the reference)")
    big = float(w.abs().max()) if w.numel() else 0.0
    if big > 65504.0:
        raise ValueError(f"split_f16: |value| up to {big:g} does not fit the f16 range (+-65504) of the "
                         "\"f16x3\" arithmetic; load the model with precision=\"f32\"")
    hi = w.to(torch.float16)
    lo = ((w - hi.float()) * 2048.0).to(torch.float16)
    # How well do the two planes hold THIS matrix?  hi + lo * 2^-11 carries 22 mantissa bits while
    # |w| >= 2^-14; below, the f16 subnormal spacing leaves an ABSOLUTE error of 2^-36 per element
    # (tests/test_gpu_kernels.py::test_f16x3_dynamic_range_map: fp32-grade for magnitudes 2^-13 .. 2^15,
    # 4x worse per factor 4 below).  A layer whose energy sits in such tiny weights (nothing in the
    # published architectures does; a checkpoint with, say, a BatchNorm scale of 1e6 folded elsewhere
    # could) is measured here, once, at pack time: relative representation error of the layer, RMS.
    if w.numel():
        rep = (hi.double() + lo.double() / 2048.0 - w.double()).norm() / max(float(w.double().norm()), 1e-300)
        rep = float(rep)
        SPLIT_REPORT.append((name or f"matrix{len(SPLIT_REPORT)}", tuple(w.shape), rep))
        if rep > SPLIT_LIMIT:
            from .config import setting
            msg = (f"split_f16({name or 'matrix'}): the f16x3 planes represent this layer to {rep:.2e} (relative, RMS) — "
                   f"an f32 copy rounds to ~3e-8; its weights lie below the range the split holds to 22 bits "
                   f"(|w| >= 2^-14).  Load the model with precision=\"f32\"")
            if str(setting("split_strict", None, "1")) != "0":
                raise ValueError(msg + " (DZ_ENGINE=split_strict=0 turns this into a warning)")
            import warnings
            warnings.warn(msg)
    return torch.stack([hi, lo]).view(torch.int16).contiguous()


def kb_major(planes: torch.Tensor) -> torch.Tensor:
    """``[2][R][K]`` planes (``split_f16``) -> the same elements in the "kb-major" order the LDS-DMA kernels
    read (``k_gemm_pre.hip``, ``k_mlp_head.hip``; ``dz_kb`` in ``csrc/dz_common.h``): ``[2][K / 32][R][32]``,
    i.e. the 32-wide k-tile of consecutive rows is contiguous (16 rows = one 1 KiB LDS-DMA piece = 8 full
    cache lines; row-major planes made every piece 16 half lines).  ``K`` must be a multiple of 32."""
    two, rows, k = planes.shape
    assert two == 2 and k % 32 == 0, planes.shape
    return planes.reshape(2, rows, k // 32, 32).permute(0, 2, 1, 3).contiguous()


def from_kb(planes: torch.Tensor, rows: int, k: int) -> torch.Tensor:
    """Inverse of ``kb_major``: any tensor holding ``2 * rows * k`` kb-major elements -> ``[2][rows][k]``."""
    return planes.reshape(2, k // 32, rows, 32).permute(0, 2, 1, 3).reshape(2, rows, k).contiguous()


# (name, shape, relative RMS representation error) of every matrix split so far in this process, and
# the error above which a layer is refused: 2^-20 = 9.5e-7 is ~30x an f32 rounding and about where the
# whole-network gates of this package (segmentation 1e-4 abs, embedding 1e-4 rel) would start to notice
SPLIT_REPORT: list = []
SPLIT_LIMIT = 2.0 ** -20
# largest |w| of a kb-major weight matrix (the layers on k_gemm_pre.hip / k_gemm_g2.hip): 2^11 |w| must stay an f16
KB_WEIGHT_LIMIT = 31.98


def dft_matrices(n_fft: int = 400) -> torch.Tensor:
    """The windowed DFT of ECAPA's Fbank as ONE real GEMM operand, (2 * (n_fft // 2 + 1), n_fft) f64: rows
    0 .. 200 = cos(2 pi k n / N) * w[n], rows 201 .. 401 = sin(...) * w[n] (periodic Hamming window, like
    torch.stft's callers): frame @ rows.T = (Re, -Im) of rfft(frame * w); the power spectrum squares and adds
    the halves (pinned against numpy.fft by the DSP pin tests under tests/)."""
    n = torch.arange(n_fft, dtype=torch.float64)
    win = torch.hamming_window(n_fft, dtype=torch.float64)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None]
    ang = 2.0 * math.pi * k * n[None, :] / n_fft
    return torch.cat([torch.cos(ang) * win, torch.sin(ang) * win], 0)


def lstm_whh_planes(whh: torch.Tensor, variant: int) -> torch.Tensor:
    """W_hh ``[2 dir][512][128]`` (PyTorch row order gate*128 + unit) -> int16 ``[2 dir][2 planes][512][128]``
    of f16 bit patterns for the matrix-core recurrence (``k_lstm_mfma.hip``).

    variant 0: ``split_f16`` per direction (lo scaled by 2^11, two accumulators in the kernel).
    variant 1 / 2: the activation scale of each gate row is folded in, ``W' = W * s * 2^SH`` with
    ``s = -log2(e)`` (i, f, o) or ``-2 log2(e)`` (g) and ``SH = 0 / 8``; ``hi = f16(W')``,
    ``lo = f16(W' - hi)`` UNSCALED, so that one accumulator holds ``hi.hi + hi.lo + lo.hi``."""
    whh = whh.detach().float().cpu()
    assert whh.shape == (2, 512, 128), whh.shape
    if variant in (0, 3):       # variant 3 = variant 0's planes, x-projection fetched by LDS-DMA
        return torch.stack([split_f16(whh[0]), split_f16(whh[1])]).contiguous()
    if variant >= 4:            # (> 4: timing-only forms of the experiments build, same planes)
        # the software-pipelined kernel: every gate row carries its activation scale (LSTM_GATE_SCALE: an accumulator
        # is then the exp2 argument), and the contraction index is re-ordered so that each half of K holds two of a
        # lane's four cells: column k' = 64 a + 2 p + e  <-  hidden unit u = 4 p + 2 a + e
        scale = torch.tensor(LSTM_GATE_SCALE, dtype=torch.float64).view(1, 4, 1, 1)
        w = (whh.double().view(2, 4, 128, 128) * scale).float().view(2, 512, 128)
        w = w[:, :, lstm_k_order()].contiguous()
        return torch.stack([split_f16(w[0], "lstm.weight_hh (scaled)"), split_f16(w[1], "lstm.weight_hh (scaled)")]).contiguous()
    sh = {1: 0, 2: 8}[variant]
    log2e = 1.44269504088896341
    scale = torch.full((4, 1, 1), -log2e, dtype=torch.float64)
    scale[2] = -2.0 * log2e                                   # PyTorch gate order i, f, g, o
    w = (whh.double().view(2, 4, 128, 128) * scale[None] * float(2 ** sh)).float().view(2, 512, 128)
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return torch.stack([hi, lo], dim=1).view(torch.int16).contiguous()


# variant 4: exp(-x) = exp2(LSTM_GATE_SCALE x) for the gates i, f, o (sigmoid) and exp(-2x) for g (tanh), PyTorch's gate
# order i, f, g, o; folded into W_ih, the biases and W_hh of a variant-4 engine
LSTM_GATE_SCALE = (-1.44269504088896341, -1.44269504088896341, -2.88539008177792681, -1.44269504088896341)


def lstm_k_order() -> torch.Tensor:
    """hidden unit held by column k' of a variant-4 ``W_hh`` plane (and of ``h_t`` in the kernel's LDS)."""
    k = torch.arange(128)
    return 4 * ((k & 63) >> 1) + 2 * (k >> 6) + (k & 1)


def lstm_scale_gx(t: torch.Tensor, unit_major: bool = True) -> torch.Tensor:
    """rows of a stacked ``[1024][...]`` x-projection operand (weights or bias; row = dir*512 + unit*4 + gate when
    ``unit_major``, else dir*512 + gate*128 + unit) times the activation scale of their gate (variant 4)."""
    sc = torch.tensor(LSTM_GATE_SCALE, dtype=torch.float64)
    rows = torch.arange(t.shape[0])
    g = (rows & 3) if unit_major else ((rows % 512) // 128)
    shape = (-1,) + (1,) * (t.dim() - 1)
    return (t.double() * sc[g].view(shape)).float()


THROUGHPUT_LSTM_VARIANT = 4      # k_lstm_mfma.hip, the software-pipelined form: what a throughput engine runs unless told otherwise
RECURRENCES = ("valu", "0", "1", "2", "3", "4")


def lstm_variant_of(recurrence) -> int:
    """``recurrence`` ("valu" | "0" | "3" | "4"; "1" / "2": experiments build) -> -1 (one chain per CU on the f32
    vector units, k_lstm.hip) or the matrix-core variant of k_lstm_mfma.hip (``lstm_whh_planes``)."""
    r = "valu" if recurrence is None else str(recurrence)
    if r not in RECURRENCES:
        raise ValueError(f"recurrence={recurrence!r}: expected one of {RECURRENCES}")
    v = -1 if r == "valu" else int(r)
    if v in (1, 2) and not _lib.EXPERIMENTS:
        raise ValueError(f"recurrence={r}: matrix-core recurrence variants 1 / 2 exist in the experiments build only "
                         "(DZ_EXPERIMENTS=1); the shipped library has valu, 0, 3 and 4")
    return v


def _pad2(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    out = torch.zeros(rows, cols, dtype=torch.float32)
    out[: w.shape[0], : w.shape[1]] = w
    return out


def _pad1(v: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros(n, dtype=torch.float32)
    out[: v.shape[0]] = v
    return out


def _conv_pack(w: torch.Tensor, cin_pad: int, n_pad: int, k_pad: int) -> torch.Tensor:
    """Conv1d weight [co][ci][tap] -> [n_pad][k_pad] with k = tap*cin_pad + ci."""
    co, ci, taps = w.shape
    t = torch.zeros(co, taps, cin_pad, dtype=torch.float32)
    t[:, :, :ci] = w.float().permute(0, 2, 1)
    return _pad2(t.reshape(co, taps * cin_pad), n_pad, k_pad)


class _Packed:
    """Keeps the device tensors alive and exposes their addresses."""

    def __init__(self, device: torch.device):
        self.device = device
        self.tensors: List[torch.Tensor] = []

    def put(self, t: torch.Tensor) -> int:
        d = t.detach().to(dtype=torch.float32).contiguous().to(self.device)
        self.tensors.append(d)
        return d.data_ptr()

    def put_split(self, t: torch.Tensor, name: Optional[str] = None, kb: bool = False) -> int:
        """The matrix as two f16 planes (hi, lo * 2^11) for the split-f16 GEMM path; ``kb``: in the kb-major
        order of the layers that run on ``k_gemm_pre.hip`` / ``k_mlp_head.hip`` (``kb_major``)."""
        d = split_f16(t, name)
        # (only the experiments build's generation 2 / 3 GEMM has this limit: the shipped k_gemm_pre.hip keeps
        # two accumulators and takes any weight split_f16 accepts)
        if kb and _lib.exp_env("DZ_GEMM_GEN", "1") in ("2", "3") and t.numel() and float(t.detach().abs().max()) >= KB_WEIGHT_LIMIT:
            # k_gemm_g2.hip multiplies the hi plane of a weight fragment by 2^11 in f16 (one accumulator per
            # fragment): exact while |w| < 32, infinite beyond
            raise ValueError(f"{name or 'matrix'}: |weight| up to {float(t.detach().abs().max()):g} >= {KB_WEIGHT_LIMIT:g} "
                             "cannot be scaled by 2^11 inside the f16 range (k_gemm_g2.hip); load the model with "
                             "precision=\"f32\"")
        d = (kb_major(d) if kb else d).to(self.device)
        self.tensors.append(d)
        return d.data_ptr()

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors)


def _pack_sincnet(sd: Dict[str, torch.Tensor], pk: _Packed, prefix: str = "sincnet.",
                  split: bool = False) -> _lib.SincNetWeights:
    g = lambda k: sd[prefix + k].detach().cpu()
    filt = sinc_filters(g("conv1d.0.filterbank.low_hz_"), g("conv1d.0.filterbank.band_hz_"),
                        g("conv1d.0.filterbank.window_"), g("conv1d.0.filterbank.n_"))
    assert filt.shape == (80, 251)
    w = _lib.SincNetWeights()
    w.wav_gamma = float(g("wav_norm1d.weight").reshape(-1)[0])
    w.wav_beta = float(g("wav_norm1d.bias").reshape(-1)[0])
    w.filt = pk.put(fold_sinc_filters(filt))
    import os
    if split and _lib.exp_env("DZ_CONV0_SPLIT", "1") != "0":
        # the unfolded bank, zero padded to [96][256], as f16 planes for the matrix-core kernel
        w.filt_split = pk.put_split(_pad2(filt, 96, 256), prefix + "sinc filter bank")
    w.in0_g, w.in0_b = pk.put(g("norm1d.0.weight")), pk.put(g("norm1d.0.bias"))
    w.w1 = pk.put(_conv_pack(g("conv1d.1.weight"), 80, 64, 416))
    if split:
        w.w1_split = pk.put_split(_conv_pack(g("conv1d.1.weight"), 80, 64, 416), prefix + "conv1d.1")
        w.w2_split = pk.put_split(_conv_pack(g("conv1d.2.weight"), 64, 64, 320), prefix + "conv1d.2")
    w.b1 = pk.put(_pad1(g("conv1d.1.bias"), 64))
    w.in1_g, w.in1_b = pk.put(_pad1(g("norm1d.1.weight"), 64)), pk.put(_pad1(g("norm1d.1.bias"), 64))
    w.w2 = pk.put(_conv_pack(g("conv1d.2.weight"), 64, 64, 320))
    w.b2 = pk.put(_pad1(g("conv1d.2.bias"), 64))
    w.in2_g, w.in2_b = pk.put(_pad1(g("norm1d.2.weight"), 64)), pk.put(_pad1(g("norm1d.2.bias"), 64))
    return w


class PackedConv0Pair:
    """Operands of ``dz_sinc_conv0_pair`` (csrc/k_front.hip): the sinc banks of the segmentation and the embedding
    network as ONE f16x3 bank of 4 x 48 slots — wave ``w`` of the kernel owns slots ``48 w .. 48 w + 47``, of which
    the first 40 hold filters ``40 w .. 40 w + 39`` of (seg 0..79 | emb 0..79) and the rest are zero — plus, per
    slot, ``beta_net * sum_k filt[k]``: with xh the un-affine InstanceNorm1d(1) of the window,
    ``conv(gamma xh + beta) = gamma conv(xh) + beta sum(filt)``, so one split of ``xh`` serves both networks."""

    def __init__(self, seg_sd: Dict[str, torch.Tensor], emb_sd: Dict[str, torch.Tensor], device: torch.device,
                 seg_prefix: str = "sincnet.", emb_prefix: str = "sincnet."):
        banks, betas = [], []
        for sd, prefix in ((seg_sd, seg_prefix), (emb_sd, emb_prefix)):
            g = lambda k: sd[prefix + k].detach().cpu()
            filt = sinc_filters(g("conv1d.0.filterbank.low_hz_"), g("conv1d.0.filterbank.band_hz_"),
                                g("conv1d.0.filterbank.window_"), g("conv1d.0.filterbank.n_"))
            assert filt.shape == (80, 251)
            banks.append(filt)
            betas.append(float(g("wav_norm1d.bias").reshape(-1)[0]))
        full = torch.cat(banks, 0)                                   # (160, 251): seg | emb
        slots = torch.zeros(192, 256, dtype=torch.float32)
        bsum = torch.zeros(192, dtype=torch.float64)
        for w in range(4):
            slots[48 * w: 48 * w + 40, :251] = full[40 * w: 40 * w + 40]
            bsum[48 * w: 48 * w + 40] = betas[w // 2] * full[40 * w: 40 * w + 40].double().sum(1)
        self.planes = split_f16(slots, "sinc filter bank (pair)").to(device)      # int16 [2? see split_f16][192][256]
        self.bsum = bsum.float().to(device)


class PackedSegmentation:
    """``dz_seg_weights`` + the tensors behind it.

    ``struct`` is what the synchronous blocks API runs: the recurrence named by ``recurrence`` ("valu" by default: one
    chain per CU, the shortest layer).  ``struct_for(r)`` is the same network with another recurrence kernel (built on
    first use, cached): the W_hh planes of that matrix-core variant and, for variant 4, an x-projection whose rows carry
    the gates' activation scales; everything else is shared, not copied.  ``struct_throughput`` = what a throughput
    engine (``StreamBatch`` with >= 64 streams per step, several steps in flight) creates its handles from."""

    def __init__(self, sd: Dict[str, torch.Tensor], device: torch.device, powerset: bool = False,
                 num_speakers: int | None = None, precision: str = "f32", recurrence: Optional[str] = None):
        assert precision in PRECISIONS, precision
        split = precision == "f16x3"
        pk = _Packed(device)
        g = lambda k: sd[k].detach().cpu().float()
        w = _lib.SegWeights()
        w.sinc = _pack_sincnet(sd, pk, split=split)
        self._split, self._lstm, self._wih0_kb = split, [], {}
        for layer in range(4):
            # rows of the stacked W_ih (and the bias) go unit-major, dir*512 + unit*4 + gate, so the
            # x-projection GEMM writes the four gates of a unit next to each other and the recurrence
            # reads them as one 16-byte word (PyTorch's order is dir*512 + gate*128 + unit)
            um = lambda t: t.reshape(2, 4, 128, *t.shape[1:]).transpose(1, 2).reshape(t.shape).contiguous()
            wih = um(torch.cat([g(f"lstm.weight_ih_l{layer}"), g(f"lstm.weight_ih_l{layer}_reverse")], 0))
            bias = um(torch.cat([g(f"lstm.bias_ih_l{layer}") + g(f"lstm.bias_hh_l{layer}"),
                                 g(f"lstm.bias_ih_l{layer}_reverse") + g(f"lstm.bias_hh_l{layer}_reverse")], 0))
            whh = torch.stack([g(f"lstm.weight_hh_l{layer}"), g(f"lstm.weight_hh_l{layer}_reverse")], 0)
            assert whh.shape == (2, 512, 128)
            self._lstm.append((wih, bias, whh))
            w.wih[layer], sp, w.bih[layer] = self._put_proj(pk, layer, wih, bias, "")
            if split:
                w.wih_split[layer] = sp
            w.whh[layer] = pk.put(whh)
        w.lin0_w, w.lin0_b = pk.put(g("linear.0.weight")), pk.put(g("linear.0.bias"))
        w.lin1_w, w.lin1_b = pk.put(g("linear.1.weight")), pk.put(g("linear.1.bias"))
        if split:
            w.lin0_split = pk.put_split(g("linear.0.weight"), "linear.0", kb=True)
            w.lin1_split = pk.put_split(g("linear.1.weight"), "linear.1", kb=True)
        cls_w, cls_b = g("classifier.weight"), g("classifier.bias")
        ncls = cls_w.shape[0]
        w.cls_w, w.cls_b = pk.put(_pad2(cls_w, 64, 128)), pk.put(_pad1(cls_b, 64))
        w.num_classes = ncls
        w.powerset = 1 if powerset else 0
        if powerset:
            # classes = 1 + S + S(S-1)/2  ->  S
            s = num_speakers or int(round((-1 + math.sqrt(1 + 8 * (ncls - 1))) / 2))
            w.num_speakers = s
        else:
            w.num_speakers = ncls
        if split:
            w.wih0_split_kb = self._wih0_kb[""]
        self.pack = pk
        self.num_speakers = int(w.num_speakers)
        self._structs = {"valu": w}
        self.recurrence = "valu" if recurrence is None else str(recurrence)
        self.struct = self.struct_for(self.recurrence)

    def _put_proj(self, pk, layer, wm, bv, tag):
        """-> (f32 W_ih, its split planes or None, bias) of one layer on the device"""
        kpad = 64 if layer == 0 else 256
        # layer 0 runs on k_gemm_split.hip (row-major planes), layers 1..3 on k_gemm_pre.hip (kb-major)
        sp = pk.put_split(_pad2(wm, 1024, kpad), f"lstm.weight_ih_l{layer}{tag}", kb=layer > 0) if self._split else None
        if layer == 0 and self._split:      # ... and kb-major too: the first projection behind the norm + split pass (round 6)
            self._wih0_kb[tag] = pk.put_split(_pad2(wm, 1024, kpad), f"lstm.weight_ih_l0{tag} (kb)", kb=True)
        return pk.put(_pad2(wm, 1024, kpad)), sp, pk.put(bv)

    def struct_for(self, recurrence: Optional[str]):
        """The weight struct whose recurrence runs on ``recurrence`` ("valu" | "0" | "3" | "4"); exact f32 has only
        "valu" (the matrix-core kernels are split-f16 arithmetic)."""
        r = "valu" if recurrence is None else str(recurrence)
        v = lstm_variant_of(r)
        if not self._split:
            v, r = -1, "valu"
        got = self._structs.get(r)
        if got is None:
            got = _lib.SegWeights.from_buffer_copy(self._structs["valu"])
            for layer, (wih, bias, whh) in enumerate(self._lstm):
                d = lstm_whh_planes(whh, v).to(self.pack.device)        # [dir][plane][512][128] f16
                self.pack.tensors.append(d)
                got.whh_split[layer] = d.data_ptr()
                if v == 4:        # its x-projection carries the gates' activation scales
                    got.wih[layer], got.wih_split[layer], got.bih[layer] = self._put_proj(
                        self.pack, layer, lstm_scale_gx(wih), lstm_scale_gx(bias), " (scaled)")
                    if layer == 0:
                        got.wih0_split_kb = self._wih0_kb[" (scaled)"]
            got.lstm_variant = v
            self._structs[r] = got
        return got

    @property
    def struct_throughput(self):
        """A THROUGHPUT engine's weights (StreamBatch with many streams per launch and several steps in flight): the
        recurrence on the matrix cores, 16 chains per workgroup — a seventh of the CU-time of the one-chain-per-CU
        kernel at twice its latency — unless the model was built with an explicit recurrence."""
        if not self._split or self.recurrence != "valu":
            return self.struct
        return self.struct_for(str(THROUGHPUT_LSTM_VARIANT))


class PackedEmbedding:
    """``dz_emb_weights`` + the tensors behind it."""

    TDNN = [(64, 512, 512), (512, 512, 512), (512, 512, 512), (512, 512, 512), (512, 1500, 1536)]

    def __init__(self, sd: Dict[str, torch.Tensor], device: torch.device, precision: str = "f32",
                 weight_interp: str = "linear"):
        """``weight_interp``: how StatsPool resamples the pooling weights to the feature frames — "linear"
        (``F.interpolate(mode="linear")``, pyannote.audio 2.x .. 3.0) or "nearest" (pyannote.audio >= 3.1)."""
        assert precision in PRECISIONS, precision
        if weight_interp not in ("linear", "nearest"):
            raise ValueError(f"weight_interp={weight_interp!r}: expected 'linear' or 'nearest'")
        split = precision == "f16x3"
        pk = _Packed(device)
        g = lambda k: sd[k].detach().cpu().float()
        w = _lib.EmbWeights()
        w.pool_nearest = int(weight_interp == "nearest")
        w.sinc = _pack_sincnet(sd, pk, split=split)
        for i, (cin_pad, cout, npad) in enumerate(self.TDNN):
            cw = g(f"tdnns.{3 * i}.weight")
            assert cw.shape[0] == cout
            k = cw.shape[2] * cin_pad
            w.tw[i] = pk.put(_conv_pack(cw, cin_pad, npad, (k + 31) // 32 * 32))
            if split:
                # tdnn1 runs on k_gemm_split.hip (row-major planes), tdnn2..5 on k_gemm_pre.hip (kb-major)
                w.tw_split[i] = pk.put_split(_conv_pack(cw, cin_pad, npad, (k + 31) // 32 * 32), f"tdnn{i + 1}", kb=i > 0)
                if i == 0:      # ... and kb-major too: tdnn1 behind the norm + split pass (round 6)
                    w.tw0_split_kb = pk.put_split(_conv_pack(cw, cin_pad, npad, (k + 31) // 32 * 32), "tdnn1 (kb)", kb=True)
            w.tb[i] = pk.put(_pad1(g(f"tdnns.{3 * i}.bias"), npad))
            bn = f"tdnns.{3 * i + 2}."
            scale = g(bn + "weight") / torch.sqrt(g(bn + "running_var") + BN_EPS)
            shift = g(bn + "bias") - g(bn + "running_mean") * scale
            w.ts[i], w.th[i] = pk.put(_pad1(scale, npad)), pk.put(_pad1(shift, npad))
        ew = g("embedding.weight")
        assert ew.shape == (512, 3000), "only the 512-d x-vector head is built"
        w.emb_w, w.emb_b = pk.put(_pad2(ew, 512, 3008)), pk.put(g("embedding.bias"))
        w.dimension = 512
        self.struct, self.pack = w, pk


class PackedEcapa:
    """``dz_ecapa_weights`` + the tensors behind it (speechbrain ECAPA_TDNN checkpoint keys:
    ``blocks.0.conv.conv.weight`` ... ``fc.conv.weight``).

    * the Hamming window is folded into the DFT matrix, so the STFT is one GEMM over the
      overlapping 400-sample rows of the signal (hop 160);
    * BatchNorm1d (eval) after ReLU is folded to scale / shift; ``asp_bn`` is folded into ``fc``;
    * the 9216-wide attention TDNN is split into the 3072 columns that see x and the 6144 columns
      that see the per-row global (mean | std), which become a per-row bias."""

    def __init__(self, sd: Dict[str, torch.Tensor], device: torch.device, precision: str = "f32"):
        assert precision in PRECISIONS, precision
        split = precision == "f16x3"
        pk = _Packed(device)
        g = lambda k: sd[k].detach().cpu().float()
        w = _lib.EcapaWeights()

        def bn(prefix, npad):
            scale = g(prefix + ".norm.weight") / torch.sqrt(g(prefix + ".norm.running_var") + BN_EPS)
            shift = g(prefix + ".norm.bias") - g(prefix + ".norm.running_mean") * scale
            return _pad1(scale, npad), _pad1(shift, npad)

        def layer(dst, prefix, cin_pad, npad, kpad, norm=True, weight=None, wide=False, kb=False):
            cw = g(prefix + ".conv.weight") if weight is None else weight
            dst.w = pk.put(_conv_pack(cw, cin_pad, npad, kpad))
            if wide and split:   # also as split-f16 planes: the layer runs on k_gemm_split.hip, or (kb: planes in
                #                  kb-major order) with pre-split activations on k_gemm_pre.hip
                dst.wsplit = pk.put_split(_conv_pack(cw, cin_pad, npad, kpad), prefix, kb=kb)
            dst.b = pk.put(_pad1(g(prefix + ".conv.bias"), npad))
            if norm:
                sc, sh = bn(prefix.rsplit(".conv", 1)[0] + ".norm", npad)
                dst.s, dst.h = pk.put(sc), pk.put(sh)

        # ---- features ------------------------------------------------------------------
        w.dft = pk.put(_pad2(dft_matrices().float(), 448, 416))
        if split:
            w.dft_split = pk.put_split(_pad2(dft_matrices().float(), 512, 416), "windowed DFT")
        w.mel = pk.put(_pad2(ecapa_mel_filterbank().t().contiguous(), 128, 224))
        # ---- network -------------------------------------------------------------------
        layer(w.block0, "blocks.0.conv", 80, 1024, 416, wide=True)
        for i in range(3):
            p, b = f"blocks.{i + 1}", w.ser[i]
            layer(b.tdnn1, p + ".tdnn1.conv", 1024, 1024, 1024, wide=True, kb=True)
            for j in range(7):
                layer(b.res[j], p + f".res2net_block.blocks.{j}.conv", 128, 128, 384, wide=True)
            layer(b.tdnn2, p + ".tdnn2.conv", 1024, 1024, 1024, wide=True, kb=True)
            layer(b.se1, p + ".se_block.conv1", 1024, 128, 1024, norm=False)
            layer(b.se2, p + ".se_block.conv2", 128, 1024, 128, norm=False)
        layer(w.mfa, "mfa.conv", 3072, 3072, 3072, wide=True, kb=True)
        aw = g("asp.tdnn.conv.conv.weight")                           # (128, 9216, 1)
        layer(w.asp_tdnn, "asp.tdnn.conv", 3072, 128, 3072, weight=aw[:, :3072], wide=True)
        w.asp_wms = pk.put(aw[:, 3072:, 0].contiguous())             # (128, 6144)
        layer(w.asp_conv, "asp.conv", 128, 3072, 128, norm=False, wide=True)
        sc, sh = bn("asp_bn", 6144)
        fw, fb = g("fc.conv.weight")[:, :, 0], g("fc.conv.bias")     # (192, 6144)
        w.fc.w = pk.put((fw * sc[None, :]).contiguous())
        w.fc.b = pk.put(fb + fw @ sh)
        w.zeros = pk.put(torch.zeros(6144))
        self.struct, self.pack = w, pk


def ecapa_mel_filterbank(n_mels: int = 80, n_fft: int = 400, sample_rate: int = 16000) -> torch.Tensor:
    """speechbrain Filterbank (triangular, f_min = 0, f_max = sr / 2): (n_fft // 2 + 1, n_mels)."""
    to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
    mel = torch.linspace(to_mel(0.0), to_mel(sample_rate / 2), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    all_freqs = torch.linspace(0, sample_rate // 2, n_fft // 2 + 1)
    slope = (all_freqs.repeat(n_mels, 1) - f_central[:, None]) / band[:, None]
    return torch.max(torch.zeros(1), torch.min(slope + 1.0, -slope + 1.0)).t().contiguous()
