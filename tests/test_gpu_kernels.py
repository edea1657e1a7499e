"""Per-kernel parity: every HIP kernel vs a plain PyTorch fp32 (CPU) restatement of the same op,
called through the C ABI (dz_k_* entry points).  Tolerances are fp32 round-off class: the
kernels compute in exact-f32 MFMA / f32 VALU, only the summation order differs from ATen.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from diart_amd import _lib

pytestmark = pytest.mark.gpu


def _ctx(dev):
    return _lib.context(dev.index or 0)


def _sync():
    torch.cuda.synchronize()


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# --------------------------------------------------------------------------- #
def test_wave_stats(gpu):
    g = torch.Generator().manual_seed(0)
    S = 80000
    x = torch.randn(5, S, generator=g) * 0.1 + torch.tensor([0.0, 0.3, -0.2, 0.01, 1.0])[:, None]
    big = torch.zeros(5, S + 64)
    big[:, :S] = x
    d = big.to(gpu)
    st = torch.empty(5, 2, device=gpu)
    lib = _lib.load()
    _lib.check(lib.dz_k_wave_stats(_ctx(gpu), d.data_ptr(), d.stride(0), 5, S, st.data_ptr(), None))
    _sync()
    mean = x.double().mean(1)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5)
    assert torch.allclose(st[:, 0].cpu().double(), mean, atol=1e-6)
    assert torch.allclose(st[:, 1].cpu().double(), rstd, rtol=2e-6)


def test_sinc_conv0(gpu):
    from diart_amd.synth import synth_segmentation_state, synth_stream, sliding_chunks
    from diart_amd.weights import fold_sinc_filters, sinc_filters
    sd = synth_segmentation_state()
    p = "sincnet.conv1d.0.filterbank."
    filt = sinc_filters(sd[p + "low_hz_"], sd[p + "band_hz_"], sd[p + "window_"], sd[p + "n_"])
    S, B = 80000, 3
    x = torch.from_numpy(sliding_chunks(synth_stream(5, 8.0))[:B].copy())
    gamma, beta = 1.3, -0.05
    xn = F.instance_norm(x[:, None, :]) * gamma + beta
    ref = F.max_pool1d(F.conv1d(xn, filt[:, None, :], stride=10).abs(), 3, 3)  # (B,80,2658)
    P0 = ref.shape[2]
    assert P0 == 2658
    lib = _lib.load()
    d = x.to(gpu)
    st = torch.empty(B, 2, device=gpu)
    _lib.check(lib.dz_k_wave_stats(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), None))
    fk = fold_sinc_filters(filt).to(gpu)
    nt = (7975 + 191) // 192
    y0 = torch.full((B, P0, 80), float("nan"), device=gpu)
    part = torch.full((B, nt, 80, 2), float("nan"), device=gpu)
    _lib.check(lib.dz_k_sinc_conv0(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), gamma,
                                   beta, fk.data_ptr(), y0.data_ptr(), part.data_ptr(), None))
    _sync()
    got = y0.cpu().permute(0, 2, 1)
    assert not torch.isnan(got).any()
    assert _rel(got, ref) < 2e-5
    ps = part.cpu().double().sum(1)  # (B,80,2)
    assert torch.allclose(ps[..., 0], ref.double().sum(2), rtol=1e-5)
    assert torch.allclose(ps[..., 1], (ref.double() ** 2).sum(2), rtol=1e-5)
    # finalize -> scale/shift of InstanceNorm1d(80, affine)
    gam, bet = torch.rand(80) + 0.5, torch.randn(80) * 0.1
    sc, sh = torch.empty(B, 80, device=gpu), torch.empty(B, 80, device=gpu)
    dg, db = gam.to(gpu), bet.to(gpu)
    _lib.check(lib.dz_k_finalize_norm(_ctx(gpu), part.data_ptr(), B, nt, 80, P0, dg.data_ptr(),
                                      db.data_ptr(), sc.data_ptr(), sh.data_ptr(), None))
    _sync()
    normed = got * sc.cpu()[:, :, None] + sh.cpu()[:, :, None]
    refn = F.instance_norm(ref) * gam[None, :, None] + bet[None, :, None]
    assert (normed - refn).abs().max().item() < 2e-4


@pytest.mark.parametrize("S", [80000, 40000, 2661])
def test_sinc_conv0_split(gpu, S):
    """The f16 matrix-core form of sinc_conv0 (unfolded bank, split operands, four shifted sample
    copies, pooling by in-register maximum over three interleaved frame blocks) against the same
    torch restatement and the same tolerances as the exact-f32 kernel; several window lengths
    exercise the ragged last tile."""
    from diart_amd.synth import synth_segmentation_state, synth_stream
    from diart_amd.weights import sinc_filters, split_f16, _pad2
    sd = synth_segmentation_state()
    p = "sincnet.conv1d.0.filterbank."
    filt = sinc_filters(sd[p + "low_hz_"], sd[p + "band_hz_"], sd[p + "window_"], sd[p + "n_"])
    B = 3
    wave = synth_stream(5, 8.0)
    x = torch.stack([torch.from_numpy(wave[i * 8000: i * 8000 + S].copy()) for i in range(B)])
    x[1] *= 25.0                                       # a loud chunk: the normalisation removes it
    gamma, beta = 1.3, -0.05
    xn = F.instance_norm(x[:, None, :]) * gamma + beta
    ref = F.max_pool1d(F.conv1d(xn, filt[:, None, :], stride=10).abs(), 3, 3)
    P0 = ref.shape[2]
    lib = _lib.load()
    pad = torch.zeros(B, (S + 3) // 4 * 4 + 64)
    pad[:, :S] = x
    d = pad.to(gpu)
    st = torch.empty(B, 2, device=gpu)
    _lib.check(lib.dz_k_wave_stats(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), None))
    fs = split_f16(_pad2(filt, 96, 256)).to(gpu)
    nt = lib.dz_k_conv0_split_ntile(S)
    assert nt == -(-((S - 251) // 10 + 1) // 96)
    y0 = torch.full((B, P0, 80), float("nan"), device=gpu)
    part = torch.full((B, nt, 80, 2), float("nan"), device=gpu)
    _lib.check(lib.dz_k_sinc_conv0_split(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), gamma,
                                         beta, fs.data_ptr(), y0.data_ptr(), part.data_ptr(), None))
    _sync()
    got = y0.cpu().permute(0, 2, 1)
    assert not torch.isnan(got).any()
    assert _rel(got, ref) < 2e-5
    ps = part.cpu().double().sum(1)
    assert not torch.isnan(ps).any()
    assert torch.allclose(ps[..., 0], ref.double().sum(2), rtol=1e-5)
    assert torch.allclose(ps[..., 1], (ref.double() ** 2).sum(2), rtol=1e-5)


@pytest.mark.parametrize("S", [80000, 40000, 2661])
def test_sinc_conv0_pair(gpu, S):
    """Round 5: the first SincNet stage of BOTH networks in one launch (k_front.hip sinc_conv0_pair_kernel: one
    split of the un-affine normalised window, the affine pair folded into the epilogue, 4 waves x 40 filters on
    v_mfma_f32_16x16x32_f16).  Each network's y0 / partials against the torch restatement at the tolerances of the
    one-network kernel, and against that kernel itself."""
    from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream
    from diart_amd.weights import PackedConv0Pair, _pad2, sinc_filters, split_f16
    if not _lib.experiments():
        pytest.skip("sinc_conv0_pair exists in the experiments build only (correct, slower: csrc/k_front.hip)")
    seg_sd, emb_sd = synth_segmentation_state(), synth_embedding_state()
    seg_sd = dict(seg_sd); emb_sd = dict(emb_sd)
    # affine pairs that differ between the networks and are far from (1, 0)
    seg_sd["sincnet.wav_norm1d.weight"], seg_sd["sincnet.wav_norm1d.bias"] = torch.tensor([1.3]), torch.tensor([-0.05])
    emb_sd["sincnet.wav_norm1d.weight"], emb_sd["sincnet.wav_norm1d.bias"] = torch.tensor([-0.7]), torch.tensor([0.4])
    pair = PackedConv0Pair(seg_sd, emb_sd, gpu)
    B = 3
    wave = synth_stream(5, 8.0)
    x = torch.stack([torch.from_numpy(wave[i * 8000: i * 8000 + S].copy()) for i in range(B)])
    x[1] *= 25.0
    lib = _lib.load()
    pad = torch.zeros(B, (S + 3) // 4 * 4 + 64)
    pad[:, :S] = x
    d = pad.to(gpu)
    mom = torch.empty(B, lib.dz_wave_stats_floats(), device=gpu)
    _lib.check(lib.dz_wave_stats(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, mom.data_ptr(), None), "dz_wave_stats")
    st = torch.empty(B, 2, device=gpu)
    _lib.check(lib.dz_k_wave_stats(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), None))
    nt = lib.dz_k_conv0_split_ntile(S)
    P0 = ((S - 251) // 10 + 1) // 3
    y = [torch.full((B, P0, 80), float("nan"), device=gpu) for _ in range(2)]
    part = [torch.full((B, nt, 80, 2), float("nan"), device=gpu) for _ in range(2)]
    gam = [1.3, -0.7]
    bet = [-0.05, 0.4]
    _lib.check(lib.dz_k_sinc_conv0_pair(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, mom.data_ptr(), pair.planes.data_ptr(),
                                        pair.bsum.data_ptr(), gam[0], gam[1], y[0].data_ptr(), y[1].data_ptr(),
                                        part[0].data_ptr(), part[1].data_ptr(), None), "dz_k_sinc_conv0_pair")
    _sync()
    _lib.range_check(gpu.index or 0)
    for k, sd in enumerate((seg_sd, emb_sd)):
        p = "sincnet.conv1d.0.filterbank."
        filt = sinc_filters(sd[p + "low_hz_"], sd[p + "band_hz_"], sd[p + "window_"], sd[p + "n_"])
        xn = F.instance_norm(x[:, None, :].double()) * gam[k] + bet[k]
        ref = F.max_pool1d(F.conv1d(xn, filt.double()[:, None, :], stride=10).abs(), 3, 3)
        assert ref.shape[2] == P0
        got = y[k].cpu().permute(0, 2, 1)
        assert not torch.isnan(got).any(), k
        assert _rel(got, ref.float()) < 2e-5, (k, _rel(got, ref.float()))
        ps = part[k].cpu().double().sum(1)
        assert not torch.isnan(ps).any()
        assert torch.allclose(ps[..., 0], ref.sum(2), rtol=1e-5)
        assert torch.allclose(ps[..., 1], (ref ** 2).sum(2), rtol=1e-5)
        # the one-network kernel on the same inputs (it splits gamma xh + beta instead of xh: last-bits differences)
        fs = split_f16(_pad2(filt, 96, 256)).to(gpu)
        y1 = torch.full((B, P0, 80), float("nan"), device=gpu)
        p1 = torch.full((B, nt, 80, 2), float("nan"), device=gpu)
        _lib.check(lib.dz_k_sinc_conv0_split(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), gam[k], bet[k],
                                             fs.data_ptr(), y1.data_ptr(), p1.data_ptr(), None))
        _sync()
        assert _rel(y[k].cpu(), y1.cpu()) < 5e-6, (k, _rel(y[k].cpu(), y1.cpu()))


# --------------------------------------------------------------------------- #
def _run_convgemm(gpu, X, W, bias, *, taps, dil, epi, Npad, Nstore, Kpad, e0=None, e1=None,
                  nscale=None, nshift=None, Tstore=None, ldy=None, ksplit=0, split=False, rowbias=None):
    """X (B,Tin,Cin) channels-last; W (Npad,Kpad) packed.  ``split``: run the split-f16 MFMA kernel
    (dz_k_gemm_split, weights as f16 hi/lo planes) instead of the exact-f32 one."""
    lib = _lib.load()
    B, Tin, Cin = X.shape
    Tout = Tin - (taps - 1) * dil
    Tstore = Tout if Tstore is None else Tstore
    ldy = Nstore if ldy is None else ldy
    dX, dW, db = X.contiguous().to(gpu), W.contiguous().to(gpu), bias.to(gpu)
    Y = torch.full((max(ksplit, 1) * B, Tstore, ldy), float("nan"), device=gpu)
    d = _lib.ConvGemmDesc()
    d.ksplit, d.ysplit = ksplit, B * Tstore * ldy
    d.X, d.W, d.bias, d.Y = dX.data_ptr(), dW.data_ptr(), db.data_ptr(), Y.data_ptr()
    keep = [dX, dW, db]
    if e0 is not None:
        de0, de1 = e0.to(gpu), e1.to(gpu)
        d.e0, d.e1 = de0.data_ptr(), de1.data_ptr()
        keep += [de0, de1]
    if nscale is not None:
        dsc, dsh = nscale.contiguous().to(gpu), nshift.contiguous().to(gpu)
        d.nscale, d.nshift, d.nld, d.norm_on_load = dsc.data_ptr(), dsh.data_ptr(), nscale.shape[1], 1
        keep += [dsc, dsh]
    if rowbias is not None:
        drb = rowbias.contiguous().to(gpu)
        d.rowbias = drb.data_ptr()
        keep.append(drb)
    ntile = lib.dz_k_convgemm_ntile(Tout)
    part = None
    if epi == _lib.EPI_POOL3:
        part = torch.full((B, ntile, Npad, 2), float("nan"), device=gpu)
        d.partials = part.data_ptr()
    d.B, d.Tin, d.Tout, d.Cin, d.taps, d.dil = B, Tin, Tout, Cin, taps, dil
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.Tstore = taps * Cin, Kpad, Npad, Nstore, Cin, ldy, Tstore
    d.xbs, d.ybs, d.epi = Tin * Cin, Tstore * ldy, epi
    if split:
        from diart_amd.weights import split_f16
        ws = split_f16(W).to(gpu)
        d.Wsplit = ws.data_ptr()
        keep.append(ws)
        fn = lib.dz_k_conv_pool if split == "convpool" else lib.dz_k_gemm_split
        _lib.check(fn(_ctx(gpu), C.byref(d), None), "dz_k_gemm_split / dz_k_conv_pool")
    else:
        _lib.check(lib.dz_k_convgemm(_ctx(gpu), C.byref(d), None), "dz_k_convgemm")
    _sync()
    return Y.cpu(), (part.cpu() if part is not None else None)


def _pack(w, cin_pad, npad, kpad):
    from diart_amd.weights import _conv_pack
    return _conv_pack(w, cin_pad, npad, kpad)


@pytest.mark.parametrize("M,K,N,epi", [(293 * 2 + 5, 64, 1024, "bias"), (500, 256, 1024, "bias"),
                                       (97, 256, 128, "leaky"), (200, 128, 3, "sigmoid"),
                                       (192, 3008, 512, "bias64"), (192, 3008, 512, "split16"),
                                       (7, 96, 128, "split3"),
                                       (293 * 2 + 5, 64, 1024, "bias/f16x3"), (500, 256, 1024, "bias/f16x3"),
                                       (97, 256, 128, "leaky/f16x3"), (18752, 256, 1024, "bias/f16x3")])
def test_convgemm_linear(gpu, M, K, N, epi):
    epi, _, mode = epi.partition("/")
    split = mode == "f16x3"
    g = torch.Generator().manual_seed(M + K)
    X = torch.randn(1, M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    Npad = 64 if N <= 64 else (N + 127) // 128 * 128
    ksplit = int(epi[5:]) if epi.startswith("split") else 0
    if epi == "bias64" or ksplit:
        Npad = N
    Wp = torch.zeros(Npad, K)
    Wp[:N] = W
    bp = torch.zeros(Npad)
    bp[:N] = b
    code = {"bias": _lib.EPI_BIAS, "bias64": _lib.EPI_BIAS, "leaky": _lib.EPI_BIAS_LEAKY,
            "sigmoid": _lib.EPI_BIAS_SIGMOID}.get(epi, _lib.EPI_BIAS)
    Y, _ = _run_convgemm(gpu, X, Wp, bp, taps=1, dil=1, epi=code, Npad=Npad, Nstore=N, Kpad=K,
                         ksplit=ksplit, split=split)
    if ksplit:   # partial sums of the K slices, bias in slice 0
        assert not torch.isnan(Y).any()
        Y = Y.double().sum(0, keepdim=True).float()
    ref = X[0].double() @ W.double().t() + b.double()
    if epi == "leaky":
        ref = F.leaky_relu(ref, 0.01)
    if epi == "sigmoid":
        ref = torch.sigmoid(ref)
    assert not torch.isnan(Y).any()
    assert (Y[0].double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("Cin,Cout,taps,dil,Tin", [(60, 512, 5, 1, 293), (512, 512, 3, 2, 289),
                                                   (512, 512, 3, 3, 285), (512, 1500, 1, 1, 279)])
@pytest.mark.parametrize("split", [False, True], ids=["f32", "f16x3"])
def test_convgemm_tdnn(gpu, Cin, Cout, taps, dil, Tin, split):
    g = torch.Generator().manual_seed(Cin + taps)
    B = 2
    cin_pad = 64 if Cin == 60 else Cin
    x = torch.randn(B, Cin, Tin, generator=g)
    w = torch.randn(Cout, Cin, taps, generator=g) / math.sqrt(Cin * taps)
    bias = torch.randn(Cout, generator=g) * 0.1
    s0, s1 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    npad = (Cout + 127) // 128 * 128
    kpad = (taps * cin_pad + 31) // 32 * 32
    Wp = _pack(w, cin_pad, npad, kpad)
    pad1 = lambda v: torch.cat([v, torch.zeros(npad - Cout)])
    X = torch.zeros(B, Tin, cin_pad)
    X[:, :, :Cin] = x.permute(0, 2, 1)
    nscale = nshift = None
    xin = x
    if Cin == 60:  # first TDNN consumes the instance-normed SincNet output: norm-on-load
        nscale = torch.zeros(B, 64)
        nshift = torch.zeros(B, 64)
        nscale[:, :60] = torch.rand(B, 60, generator=g) + 0.5
        nshift[:, :60] = torch.randn(B, 60, generator=g) * 0.2
        xin = F.leaky_relu(x * nscale[:, :60, None] + nshift[:, :60, None], 0.01)
    Y, _ = _run_convgemm(gpu, X, Wp, pad1(bias), taps=taps, dil=dil, epi=_lib.EPI_TDNN, Npad=npad,
                         Nstore=npad, Kpad=kpad, e0=pad1(s0), e1=pad1(s1), nscale=nscale, nshift=nshift,
                         split=split)
    ref = F.leaky_relu(F.conv1d(xin.double(), w.double(), bias.double(), dilation=dil), 0.01)
    ref = ref * s0.double()[None, :, None] + s1.double()[None, :, None]
    got = Y[:, :, :Cout].permute(0, 2, 1).double()
    assert not torch.isnan(Y).any()
    assert (Y[:, :, Cout:] == 0).all()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("Cin,Tin", [(80, 2658), (60, 884), (60, 101), (80, 389)])
@pytest.mark.parametrize("split", [False, True, "convpool"], ids=["f32", "f16x3", "convpool"])
def test_convgemm_pool3(gpu, Cin, Tin, split):
    """SincNet stages 1 / 2: the exact-f32 GEMM, the split-f16 GEMM and the dedicated
    k_conv_pool.hip kernel (input tile resident in LDS, weights in registers) — same gates."""
    g = torch.Generator().manual_seed(Cin)
    B, Cout, taps = 2, 60, 5
    cin_pad = 80 if Cin == 80 else 64
    x = torch.randn(B, Cin, Tin, generator=g).abs()
    w = torch.randn(Cout, Cin, taps, generator=g) / math.sqrt(Cin * taps)
    bias = torch.randn(Cout, generator=g) * 0.1
    kpad = (taps * cin_pad + 31) // 32 * 32
    Wp = _pack(w, cin_pad, 64, kpad)
    X = torch.zeros(B, Tin, cin_pad)
    X[:, :, :Cin] = x.permute(0, 2, 1)
    nscale, nshift = torch.zeros(B, cin_pad), torch.zeros(B, cin_pad)
    nscale[:, :Cin] = torch.rand(B, Cin, generator=g) + 0.5
    nshift[:, :Cin] = torch.randn(B, Cin, generator=g) * 0.2 - 0.3
    xin = F.leaky_relu(x * nscale[:, :Cin, None] + nshift[:, :Cin, None], 0.01)
    ref = F.max_pool1d(F.conv1d(xin.double(), w.double(), bias.double()), 3, 3)
    Tp = ref.shape[2]
    bp = torch.cat([bias, torch.zeros(4)])
    Y, part = _run_convgemm(gpu, X, Wp, bp, taps=taps, dil=1, epi=_lib.EPI_POOL3, Npad=64, Nstore=64,
                            Kpad=kpad, nscale=nscale, nshift=nshift, Tstore=Tp, ldy=64, split=split)
    got = Y[:, :, :Cout].permute(0, 2, 1).double()
    assert not torch.isnan(Y).any() and not torch.isnan(part).any()
    assert (Y[:, :, Cout:] == 0).all()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    ps = part.double().sum(1)[:, :Cout]
    assert torch.allclose(ps[..., 0], ref.sum(2), rtol=1e-5, atol=1e-4)
    assert torch.allclose(ps[..., 1], (ref ** 2).sum(2), rtol=1e-5, atol=1e-4)


def test_sinc_conv0_split_more_chunks_than_a_workgroup_range_holds(gpu):
    """a workgroup of sinc_conv0_v2 keeps the statistics of the first TWO chunks of its tile range; the
    512-workgroup grid gave ranges of ntile + 2 tiles beyond ~520 chunks (reachable through the reference-shaped
    (batch x speakers) rows call with a large max_batch) and the third chunk was normalised with whatever lay behind
    the two slots.  700 short chunks (3 tiles each: 2 100 tiles, ranges of 5 on 512 workgroups), every chunk at its
    own loudness so that a borrowed (mean, rstd) shows."""
    from diart_amd.synth import synth_segmentation_state, synth_stream
    from diart_amd.weights import sinc_filters, split_f16, _pad2
    sd = synth_segmentation_state()
    p = "sincnet.conv1d.0.filterbank."
    filt = sinc_filters(sd[p + "low_hz_"], sd[p + "band_hz_"], sd[p + "window_"], sd[p + "n_"])
    B, S = 700, 2661
    wave = synth_stream(9, (B * 40 + S) / 16000.0 + 1.0)
    x = torch.stack([torch.from_numpy(wave[i * 40: i * 40 + S].copy()) for i in range(B)])
    x = x * (1.0 + (torch.arange(B) % 13).float()[:, None]) + 0.01 * (torch.arange(B) % 7).float()[:, None]
    gamma, beta = 1.1, 0.03
    xn = F.instance_norm(x[:, None, :]) * gamma + beta
    ref = F.max_pool1d(F.conv1d(xn, filt[:, None, :], stride=10).abs(), 3, 3)
    P0 = ref.shape[2]
    lib = _lib.load()
    d = torch.zeros(B, (S + 3) // 4 * 4 + 64)
    d[:, :S] = x
    d = d.to(gpu)
    st = torch.empty(B, 2, device=gpu)
    _lib.check(lib.dz_k_wave_stats(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), None))
    fs = split_f16(_pad2(filt, 96, 256)).to(gpu)
    nt = lib.dz_k_conv0_split_ntile(S)
    assert nt == 3 and -(-nt * B // 512) >= nt + 2          # the old grid's ranges spanned three chunks
    y0 = torch.full((B, P0, 80), float("nan"), device=gpu)
    part = torch.full((B, nt, 80, 2), float("nan"), device=gpu)
    _lib.check(lib.dz_k_sinc_conv0_split(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), gamma, beta,
                                         fs.data_ptr(), y0.data_ptr(), part.data_ptr(), None))
    _sync()
    got = y0.cpu().permute(0, 2, 1)
    assert not torch.isnan(got).any()
    per_chunk = ((got - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1))
    assert per_chunk.max().item() < 2e-5, (int(per_chunk.argmax()), per_chunk.max().item())


@pytest.mark.parametrize("mode", [None, "DZ_CONV0_ROT=1", "DZ_CONV0_ROT=2", "DZ_CONV0_ROT=3", "DZ_CONV0_V2=0"],
                         ids=["shipped", "no-complementary-roles", "inverted-roles", "roles-by-wave-index", "three-wave-kernel"])
def test_sinc_conv0_split_many_tiles_per_workgroup(gpu, monkeypatch, mode):
    """sinc_conv0_v2 at a batch where a persistent workgroup walks 6 - 7 tiles (40 chunks x 84 tiles on 512 workgroups):
    the double-buffered sample copies, the light waves' deferred partials, chunk changes inside a workgroup's range and
    two workgroups per CU with complementary roles.  Experiments build: the other role assignments — among them the
    fall-back by wave index, which this hardware never takes by itself — and the three-wave kernel of rounds 2 - 4
    must give the same rows (heavy-wave channels bit for bit)."""
    if mode and not _lib.experiments():
        pytest.skip("role / kernel switches exist in the experiments build only")
    from diart_amd.synth import synth_segmentation_state, synth_stream
    from diart_amd.weights import sinc_filters, split_f16, _pad2
    sd = synth_segmentation_state()
    p = "sincnet.conv1d.0.filterbank."
    filt = sinc_filters(sd[p + "low_hz_"], sd[p + "band_hz_"], sd[p + "window_"], sd[p + "n_"])
    B, S = 40, 80000
    wave = synth_stream(7, 30.0)
    x = torch.stack([torch.from_numpy(wave[i * 8000: i * 8000 + S].copy()) for i in range(B)])
    x[3] *= 25.0
    gamma, beta = 0.9, 0.02
    xn = F.instance_norm(x[:, None, :]) * gamma + beta
    ref = F.max_pool1d(F.conv1d(xn, filt[:, None, :], stride=10).abs(), 3, 3)
    P0 = ref.shape[2]
    lib = _lib.load()
    d = torch.zeros(B, S + 64)
    d[:, :S] = x
    d = d.to(gpu)
    st = torch.empty(B, 2, device=gpu)
    _lib.check(lib.dz_k_wave_stats(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), None))
    fs = split_f16(_pad2(filt, 96, 256)).to(gpu)
    nt = lib.dz_k_conv0_split_ntile(S)

    def run():
        y0 = torch.full((B, P0, 80), float("nan"), device=gpu)
        part = torch.full((B, nt, 80, 2), float("nan"), device=gpu)
        _lib.check(lib.dz_k_sinc_conv0_split(_ctx(gpu), d.data_ptr(), d.stride(0), B, S, st.data_ptr(), gamma, beta,
                                             fs.data_ptr(), y0.data_ptr(), part.data_ptr(), None))
        _sync()
        return y0.cpu(), part.cpu()

    base_y, base_p = run()
    if mode:
        k, v = mode.split("=")
        monkeypatch.setenv(k, v)
    y0, part = run()
    got = y0.permute(0, 2, 1)
    assert not torch.isnan(got).any() and not torch.isnan(part).any()
    assert _rel(got, ref) < 2e-5
    ps = part.double().sum(1)
    assert torch.allclose(ps[..., 0], ref.double().sum(2), rtol=1e-5)
    assert torch.allclose(ps[..., 1], (ref.double() ** 2).sum(2), rtol=1e-5)
    y1, part1 = run()
    assert torch.equal(y0, y1) and torch.equal(part, part1)              # deterministic
    assert torch.equal(y0[:, :, :64], base_y[:, :, :64])                   # 32x32x16 arithmetic: every variant
    assert (y0 - base_y).abs().max().item() <= 2e-6 * base_y.abs().max().item()
    assert torch.allclose(part, base_p, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("split", [False, True], ids=["f32", "f16x3"])
def test_convgemm_row_bias_relu_bn_tanh(gpu, split):
    """ECAPA's attention TDNN (speechbrain AttentiveStatisticsPooling.tdnn + tanh): 1 x 1, 3072 -> 128, a per-batch-item
    bias (the global-context columns folded into one vector per row), ReLU -> folded BatchNorm -> tanh; the exact-f32
    kernel and, since round 5, the split-f16 one (it used to fall back to the f32 kernel for this layer)."""
    g = torch.Generator().manual_seed(31)
    B, T, Cin, N = 3, 157, 3072, 128
    X = torch.randn(B, T, Cin, generator=g)
    W = torch.randn(N, Cin, generator=g) / math.sqrt(Cin)
    bias = torch.randn(N, generator=g) * 0.1
    rb = torch.randn(B, N, generator=g) * 0.3
    s0, s1 = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    Y, _ = _run_convgemm(gpu, X, W, bias, taps=1, dil=1, epi=_lib.EPI_RELU_BN_TANH, Npad=N, Nstore=N, Kpad=Cin, e0=s0, e1=s1,
                         rowbias=rb, split=split)
    ref = torch.tanh(torch.relu(X.double() @ W.double().t() + bias.double() + rb.double()[:, None, :]) * s0.double() + s1.double())
    assert not torch.isnan(Y).any()
    err = (Y.double() - ref).abs().max().item()
    print(f"row bias + tanh, {'f16x3' if split else 'f32'}: max |d| {err:.2e}")
    assert err < 1e-5            # K = 3072 products accumulated in f32; |y| <= 1


@pytest.mark.parametrize("v2", [False, True], ids=["conv_pool_h", "conv_pool_v2"])
@pytest.mark.parametrize("B,Cin,Tin", [(20, 80, 2658), (60, 60, 884), (300, 60, 101)])
def test_conv_pool_many_tiles_per_workgroup(gpu, monkeypatch, B, Cin, Tin, v2):
    """k_conv_pool.hip at batch sizes where a persistent workgroup walks SEVERAL tiles (560 - 600 tiles on 512 / 256
    workgroups): chunk changes inside a workgroup's range, the ragged last tile of every chunk and — for
    conv_pool_v2 (experiments build, DZ_CONV_POOL_V2=1) — the double-buffered input and the hand-over of finished
    blocks to the service waves.  Against the f64 layer; conv_pool_v2 also against conv_pool_h (the same products,
    added in another order)."""
    if v2 and not _lib.experiments():
        pytest.skip("conv_pool_v2 is only in the experiments build")
    g = torch.Generator().manual_seed(B + Cin)
    Cout, taps = 60, 5
    cin_pad = 80 if Cin == 80 else 64
    x = torch.randn(B, Cin, Tin, generator=g).abs()
    w = torch.randn(Cout, Cin, taps, generator=g) / math.sqrt(Cin * taps)
    bias = torch.randn(Cout, generator=g) * 0.1
    kpad = (taps * cin_pad + 31) // 32 * 32
    Wp = _pack(w, cin_pad, 64, kpad)
    X = torch.zeros(B, Tin, cin_pad)
    X[:, :, :Cin] = x.permute(0, 2, 1)
    nscale, nshift = torch.zeros(B, cin_pad), torch.zeros(B, cin_pad)
    nscale[:, :Cin] = torch.rand(B, Cin, generator=g) + 0.5
    nshift[:, :Cin] = torch.randn(B, Cin, generator=g) * 0.2 - 0.3
    xin = F.leaky_relu(x * nscale[:, :Cin, None] + nshift[:, :Cin, None], 0.01)
    ref = F.max_pool1d(F.conv1d(xin.double(), w.double(), bias.double()), 3, 3)
    Tp = ref.shape[2]
    bp = torch.cat([bias, torch.zeros(4)])
    run = lambda: _run_convgemm(gpu, X, Wp, bp, taps=taps, dil=1, epi=_lib.EPI_POOL3, Npad=64, Nstore=64, Kpad=kpad,
                                nscale=nscale, nshift=nshift, Tstore=Tp, ldy=64, split="convpool")
    if v2:
        Y0, part0 = run()
        monkeypatch.setenv("DZ_CONV_POOL_V2", "1")
    Y, part = run()
    got = Y[:, :, :Cout].permute(0, 2, 1).double()
    assert not torch.isnan(Y).any() and not torch.isnan(part).any()
    assert (Y[:, :, Cout:] == 0).all()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    ps = part.double().sum(1)[:, :Cout]
    assert torch.allclose(ps[..., 0], ref.sum(2), rtol=1e-5, atol=1e-4)
    assert torch.allclose(ps[..., 1], (ref ** 2).sum(2), rtol=1e-5, atol=1e-4)
    Y2, part2 = run()                                     # deterministic
    assert torch.equal(Y, Y2) and torch.equal(part, part2)
    if v2:
        assert (Y - Y0).abs().max().item() < 2e-6 * max(1.0, Y0.abs().max().item())
        assert torch.allclose(part, part0, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,Tin,Cin,taps,dil,N,Nstore,epi", [
    (1, 64 * 289, 512, 3, 2, 512, 512, "tdnn"),          # config-2 tdnn2, flattened
    (1, 1000, 512, 1, 1, 1536, 1500, "tdnn"),            # tdnn5: stored columns end inside a 4-column group
    (1, 18752, 256, 1, 1, 1024, 1024, "bias"),           # LSTM projection
    (3, 293, 256, 1, 1, 128, 128, "leaky"),              # MLP, batched, ragged last row tile
    (2, 131, 128, 3, 3, 256, 255, "bias"),               # taps + dilation, batched, odd Nstore
    (1, 97, 32, 1, 1, 128, 3, "sigmoid"),                # one k-tile, three stored columns
    (1, 300, 1024, 1, 1, 384, 384, "relu"), (1, 300, 1024, 1, 1, 384, 384, "relu_bn"),
    (1, 130, 96, 5, 1, 128, 128, "relu_bn_tanh")])
def test_gemm_f32_wide_layers(gpu, monkeypatch, B, Tin, Cin, taps, dil, N, Nstore, epi):
    """k_gemm_f32.hip (LDS-DMA operands, 128 x 128 tiles, v_mfma_f32_32x32x2_f32) against the f64 contraction and
    against the round-1 kernel it replaces for these layers (k_convgemm.hip, option f32_gemm = 0): the same products, one
    exact f32 FMA each, in another order."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(Tin + Cin + N)
    K = taps * Cin
    Tout = Tin - (taps - 1) * dil
    X = torch.randn(B, Tin, Cin, generator=g)
    W = torch.zeros(N, K)
    W[:Nstore] = torch.randn(Nstore, K, generator=g) / math.sqrt(K)
    bias, e0, e1 = torch.zeros(N), torch.ones(N), torch.zeros(N)
    bias[:Nstore] = torch.randn(Nstore, generator=g) * 0.1
    e0[:Nstore] = torch.rand(Nstore, generator=g) + 0.5
    e1[:Nstore] = torch.randn(Nstore, generator=g) * 0.1
    code = {"bias": _lib.EPI_BIAS, "leaky": _lib.EPI_BIAS_LEAKY, "sigmoid": _lib.EPI_BIAS_SIGMOID, "tdnn": _lib.EPI_TDNN,
            "relu": 5, "relu_bn": 6, "relu_bn_tanh": 7}[epi]
    dX, dW, db, de0, de1 = (t.contiguous().to(gpu) for t in (X, W, bias, e0, e1))
    outs = {}
    for which in ("new", "old"):
        Y = torch.full((B, Tout, Nstore), float("nan"), device=gpu)
        d = _lib.ConvGemmDesc()
        d.X, d.W, d.bias, d.Y, d.e0, d.e1 = dX.data_ptr(), dW.data_ptr(), db.data_ptr(), Y.data_ptr(), de0.data_ptr(), de1.data_ptr()
        d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = B, Tin, Tout, Tout, Cin, taps, dil
        d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy = K, K, N, Nstore, Cin, Nstore
        d.xbs, d.ybs, d.epi = Tin * Cin, Tout * Nstore, code
        if Nstore % 4:          # 16-byte stores need ldy % 4 == 0: such a layer stays on the round-1 kernel
            d.ldy = (Nstore + 3) // 4 * 4
            Y = torch.full((B, Tout, d.ldy), float("nan"), device=gpu)
            d.Y, d.ybs = Y.data_ptr(), Tout * d.ldy
        if which == "new":
            _lib.check(lib.dz_k_gemm_f32(_ctx(gpu), C.byref(d), None), "dz_k_gemm_f32")
        else:
            _lib.set_option("f32_gemm", 0)
            try:
                _lib.check(lib.dz_k_convgemm(_ctx(gpu), C.byref(d), None), "dz_k_convgemm")
            finally:
                _lib.set_option("f32_gemm", 1)
        _sync()
        outs[which] = Y[:, :, :Nstore].cpu()
        if d.ldy > Nstore:
            assert torch.isnan(Y[:, :, Nstore:]).all()           # nothing is written beyond the stored columns
    ref = torch.zeros(B, Tout, Nstore, dtype=torch.float64)
    Wd = W[:Nstore].double()
    for tp in range(taps):
        ref += X[:, tp * dil: tp * dil + Tout].double() @ Wd[:, tp * Cin:(tp + 1) * Cin].t()
    ref += bias[:Nstore].double()
    aff = lambda v: v * e0[:Nstore].double() + e1[:Nstore].double()
    ref = {"bias": lambda v: v, "leaky": lambda v: F.leaky_relu(v, 0.01), "sigmoid": torch.sigmoid,
           "tdnn": lambda v: aff(F.leaky_relu(v, 0.01)), "relu": torch.relu, "relu_bn": lambda v: aff(torch.relu(v)),
           "relu_bn_tanh": lambda v: torch.tanh(aff(torch.relu(v)))}[epi](ref)
    scale = max(1.0, ref.abs().max().item())
    for which, Y in outs.items():
        assert not torch.isnan(Y).any(), which
        assert (Y.double() - ref).abs().max().item() < 2e-5 * scale, which
    assert (outs["new"] - outs["old"]).abs().max().item() < 1e-5 * scale


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("kernel", ["valu", "mfma0", "mfma0_um", "mfma1", "mfma1_um", "mfma2", "mfma2_um", "mfma3_um", "mfma4_um"])
@pytest.mark.parametrize("B,T", [(1, 293), (3, 40), (17, 64), (32, 293), (5, 1), (2, 2), (64, 293)])
def test_lstm_recurrence(gpu, B, T, kernel):
    """k_lstm.hip (one chain per CU, exact f32) and k_lstm_mfma.hip (16 chains per workgroup on the
    f16 matrix cores, split operands; both gx column orders) against torch.nn.LSTM on the CPU —
    the SAME tolerance for both."""
    from diart_amd.weights import lstm_whh_planes
    if kernel[:5] in ("mfma1", "mfma2") and not _lib.experiments():
        pytest.skip("matrix-core recurrence variants 1 / 2 exist in the experiments build only")
    g = torch.Generator().manual_seed(B * 100 + T)
    H, I = 128, 32
    lstm = torch.nn.LSTM(I, H, 1, bidirectional=True, batch_first=True)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.25)
    x = torch.randn(B, T, I, generator=g)
    with torch.no_grad():
        ref, _ = lstm(x)
        gx = torch.cat([x @ lstm.weight_ih_l0.t() + lstm.bias_ih_l0 + lstm.bias_hh_l0,
                        x @ lstm.weight_ih_l0_reverse.t() + lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse],
                       dim=-1)  # (B,T,1024), PyTorch column order dir*512 + gate*128 + unit
        whh = torch.stack([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]).contiguous()
    hout = torch.full((B, T, 256), float("nan"), device=gpu)
    lib = _lib.load()
    if kernel == "valu":
        dgx, dw = gx.contiguous().to(gpu), whh.to(gpu)
        _lib.check(lib.dz_k_lstm(_ctx(gpu), dgx.data_ptr(), dw.data_ptr(), hout.data_ptr(), B, T, None))
    else:
        um, variant = kernel.endswith("_um"), int(kernel[4])
        if um:   # dir*512 + unit*4 + gate
            gx = gx.view(B, T, 2, 4, 128).transpose(3, 4).reshape(B, T, 1024)
        if variant >= 4:   # the software-pipelined kernel takes the x-projection times the gates' activation scales
            from diart_amd.weights import LSTM_GATE_SCALE
            gx = (gx.double().view(B, T, 256, 4) * torch.tensor(LSTM_GATE_SCALE, dtype=torch.float64)).float().view(B, T, 1024)
        dgx = gx.contiguous().to(gpu)
        dw = lstm_whh_planes(whh, variant).to(gpu)
        _lib.check(lib.dz_k_lstm_mfma(_ctx(gpu), dgx.data_ptr(), dw.data_ptr(), hout.data_ptr(), B, T,
                                      int(um), variant, None))
    _sync()
    got = hout.cpu()
    assert not torch.isnan(got).any()
    assert (got - ref).abs().max().item() < 2e-5


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("K,Fw,T", [(1, 293, 279), (3, 293, 279), (4, 293, 279), (5, 293, 279), (3, 279, 279),
                                    (1, 0, 279), (3, 293, 321), (3, 589, 571), (2, 40, 37)])
@pytest.mark.parametrize("interp", ["linear", "nearest"])
def test_stats_pool(gpu, K, Fw, T, interp):
    """interp: how the (N, Fw) weights are resampled to the T feature frames — F.interpolate(mode="linear") of
    pyannote.audio 2.x .. 3.0, or mode="nearest" of >= 3.1 (a negative weight_frames at the kernel-level entry)."""
    from oracle.models_ref import stats_pool_ref
    if interp == "nearest" and Fw in (0, T):
        pytest.skip("nothing to resample")
    g = torch.Generator().manual_seed(K * 7 + Fw)
    nx, Cc, ld = 3, 1500, 1536
    x = torch.randn(nx, T, ld, generator=g) + 0.5
    rows = nx * K
    w = None
    if Fw:
        w = torch.rand(rows, Fw, generator=g) ** 3 + 1e-8
    dx = x.to(gpu)
    dw = w.to(gpu) if w is not None else None
    out = torch.full((rows, 3008), float("nan"), device=gpu)
    _lib.check(_lib.load().dz_k_stats_pool(_ctx(gpu), dx.data_ptr(), T, Cc, ld,
                                           dw.data_ptr() if dw is not None else None,
                                           -Fw if interp == "nearest" else Fw, rows, K,
                                           out.data_ptr(), 3008, None))
    _sync()
    seq = x[:, :, :Cc].permute(0, 2, 1).repeat_interleave(K, dim=0)  # (rows,C,T)
    ref = stats_pool_ref(seq, w, interp_mode=interp)
    got = out.cpu()[:, :3000]
    assert not torch.isnan(got).any()
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("K", [3, 4])
def test_osp(gpu, K, normalize):
    from oracle.functional_ref import overlapped_speech_penalty_ref
    g = torch.Generator().manual_seed(K)
    B, Fr = 5, 293
    seg = torch.rand(B, Fr, K, generator=g)
    seg[1, :, 0] = 0.0
    seg[2] = 0.5  # constant -> min == max -> NaN -> 1e-8 when normalising
    ref = overlapped_speech_penalty_ref(seg, 3.0, 10.0, normalize)
    d = seg.to(gpu)
    for major in (0, 1):
        out = torch.empty(B, Fr, K, device=gpu)
        _lib.check(_lib.load().dz_osp(_ctx(gpu), d.data_ptr(), B, Fr, K, 3.0, 10.0, int(normalize),
                                      major, out.data_ptr(), None))
        _sync()
        got = out.cpu()
        if major:
            got = got.view(B, K, Fr).permute(0, 2, 1)
        assert torch.allclose(got, ref, rtol=2e-5, atol=1e-9)


def test_l2norm_and_cdist(gpu):
    from scipy.spatial.distance import cdist
    g = torch.Generator().manual_seed(3)
    e = torch.randn(7, 3, 512, generator=g)
    d = e.clone().to(gpu)
    _lib.check(_lib.load().dz_l2_normalize(_ctx(gpu), d.data_ptr(), 21, 512, 1.0, None))
    _sync()
    ref = e / e.norm(dim=-1, keepdim=True)
    assert torch.allclose(d.cpu(), ref, rtol=1e-5, atol=1e-7)
    cen = torch.randn(7, 20, 512, generator=g, dtype=torch.float64)
    cen[:, 5] = 0.0  # unused centroid -> NaN like scipy
    dc = cen.to(gpu)
    out = torch.empty(7, 3, 20, dtype=torch.float64, device=gpu)
    _lib.check(_lib.load().dz_cdist_cosine(_ctx(gpu), d.data_ptr(), dc.data_ptr(), 7, 3, 20, 512,
                                           out.data_ptr(), None))
    _sync()
    got = out.cpu().numpy()
    for n in range(7):
        with np.errstate(invalid="ignore", divide="ignore"):
            r = cdist(d.cpu().numpy()[n].astype(np.float64), cen.numpy()[n], metric="cosine")
        assert np.isnan(got[n][:, 5]).all()
        m = ~np.isnan(r)
        assert np.allclose(got[n][m], r[m], rtol=0, atol=1e-12)


def test_powerset(gpu):
    from oracle.models_ref import powerset_to_multilabel
    g = torch.Generator().manual_seed(9)
    lp = torch.log_softmax(torch.randn(1000, 7, generator=g), dim=-1)
    d = lp.to(gpu)
    out = torch.empty(1000, 3, device=gpu)
    _lib.check(_lib.load().dz_k_powerset(_ctx(gpu), d.data_ptr(), 1000, 7, 3, out.data_ptr(), None))
    _sync()
    assert torch.equal(out.cpu(), powerset_to_multilabel(lp))


# --------------------------------------------------------------------------- #
# pre-split path: activations travel as two f16 planes (hi, lo * 2^11)
# --------------------------------------------------------------------------- #
def _planes(x):
    """f32 (rows, cols) -> int16 (2, rows, cols) of f16 bit patterns, like weights.split_f16."""
    from diart_amd.weights import split_f16
    return split_f16(x)


def _unplanes(p):
    h = p.view(torch.float16).double()
    return h[0] + h[1] / 2048.0


def _kb(x):
    """f32 (rows, cols) -> the kb-major planes (2, cols / 32, rows, 32) k_gemm_pre.hip / k_mlp_head.hip read."""
    from diart_amd.weights import kb_major
    return kb_major(_planes(x))


def _unkb(p, rows, cols):
    """kb-major plane buffer (any shape) -> row-major (2, rows, cols)."""
    from diart_amd.weights import from_kb
    return from_kb(p, rows, cols)


@pytest.mark.parametrize("M,Cin,taps,dil,N,Nstore,epi,outs", [
    (300, 512, 3, 2, 512, 512, "tdnn", "planes"),
    (1000, 256, 1, 1, 1024, 1024, "bias", "f32"),
    (130, 128, 1, 1, 128, 128, "leaky", "both"),
    (700, 512, 3, 3, 1536, 1500, "tdnn", "both"),
    (64 * 293, 512, 1, 1, 512, 512, "tdnn", "planes"),
    (5, 256, 1, 1, 128, 128, "leaky", "f32"),
    (5157, 256, 3, 1, 512, 512, "tdnn", "both"),     # 128 x 128 tiles with a ragged last row tile
])
@pytest.mark.parametrize("kern", ["pre", "g2_mt2", "g2_mt3", "g2_mt4", "g3_mt2", "g3_mt3", "g3_mt4"])
def test_gemm_pre(gpu, M, Cin, taps, dil, N, Nstore, epi, outs, kern):
    """k_gemm_pre.hip and the three tile sizes of its generation 2, k_gemm_g2.hip (one accumulator per
    fragment, three LDS stages, counted vmcnt) and of generation 3, k_gemm_g3.hip (the same loop, persistent,
    Stream-K), against an f64 torch restatement.
    k_gemm_pre.hip (both operands as f16 hi/lo planes, tiles by LDS-DMA) against an f64 torch
    restatement: implicit-GEMM convolution over the flattened rows, every epilogue, f32 and / or
    plane output (whose hi + lo * 2^-11 must reproduce the f32 result to 2^-21), zeroed padding
    columns, ragged last tile."""
    g = torch.Generator().manual_seed(M + Cin + N)
    K = taps * Cin
    Tout = M - (taps - 1) * dil
    X = torch.randn(M, Cin, generator=g) * 1.5
    W = torch.zeros(N, K)
    W[:Nstore] = torch.randn(Nstore, K, generator=g) / math.sqrt(K)
    bias = torch.zeros(N)
    bias[:Nstore] = torch.randn(Nstore, generator=g) * 0.2
    e0, e1 = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    code = {"bias": _lib.EPI_BIAS, "leaky": _lib.EPI_BIAS_LEAKY, "tdnn": _lib.EPI_TDNN}[epi]
    dX, dW = _kb(X).to(gpu), _kb(W).to(gpu)
    db, de0, de1 = bias.to(gpu), e0.to(gpu), e1.to(gpu)
    Y = torch.full((M, N), float("nan"), device=gpu) if outs in ("f32", "both") else None
    Yp = torch.full((2, M, N), 0x7e00, dtype=torch.int16, device=gpu) if outs in ("planes", "both") else None
    d = _lib.ConvGemmDesc()
    d.Xsplit, d.xplane, d.Wsplit = dX.data_ptr(), M * Cin, dW.data_ptr()
    d.bias, d.e0, d.e1 = db.data_ptr(), de0.data_ptr(), de1.data_ptr()
    if Y is not None:
        d.Y = Y.data_ptr()
    if Yp is not None:
        d.Ysplit, d.yplane = Yp.data_ptr(), M * N
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, M, Tout, Tout, Cin, taps, dil
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = K, K, N, Nstore, Cin, N, code
    if kern != "pre" and not _lib.experiments():
        pytest.skip("generations 2 / 3 exist in the experiments build only (DZ_EXPERIMENTS=1)")
    if kern == "pre":
        _lib.check(_lib.load().dz_k_gemm_pre(_ctx(gpu), C.byref(d), None), "dz_k_gemm_pre")
    elif kern.startswith("g2"):
        _lib.check(_lib.load().dz_k_gemm_g2(_ctx(gpu), C.byref(d), int(kern[-1]), None), "dz_k_gemm_g2")
    else:       # persistent, balanced split of the (tile, k-tile) space: tiles shared by two workgroups
        _lib.check(_lib.load().dz_k_gemm_g3(_ctx(gpu), C.byref(d), int(kern[-1]), None), "dz_k_gemm_g3")
    _sync()
    _lib.range_check(gpu.index or 0)
    # reference on the operands the kernel saw (22-bit planes), f64
    Xq, Wq = _unplanes(_planes(X)), _unplanes(_planes(W))
    cols = torch.cat([Xq[j * dil: j * dil + Tout] for j in range(taps)], dim=1)   # (Tout, taps*Cin)
    ref = cols @ Wq.t() + bias.double()
    if epi == "leaky":
        ref = F.leaky_relu(ref, 0.01)
    if epi == "tdnn":
        ref = F.leaky_relu(ref, 0.01) * e0.double() + e1.double()
    scale = max(1.0, ref.abs().max().item())
    if Y is not None:
        got = Y.cpu()[:Tout, :Nstore].double()
        assert not torch.isnan(got).any()
        assert (got - ref[:, :Nstore]).abs().max().item() < 4e-6 * scale
    if Yp is not None:
        P = _unkb(Yp.cpu(), M, N)
        got = _unplanes(P)[:Tout]
        assert (got[:, :Nstore] - ref[:, :Nstore]).abs().max().item() < 4e-6 * scale
        assert (P[:, :Tout, Nstore:] == 0).all()                          # padding columns are zeros
        if Tout < M:
            assert (P[:, Tout:] == 0x7e00).all()                          # rows >= Tout untouched
        if Y is not None:   # the planes are the split of exactly the f32 output
            assert (got[:, :Nstore] - Y.cpu()[:Tout, :Nstore].double()).abs().max().item() < 2.0 ** -21 * scale


@pytest.mark.parametrize("powerset,B,F", [(False, 5, 293), (True, 3, 293), (False, 1, 37)])
def test_mlp_head(gpu, powerset, B, F):
    """k_mlp_head.hip (linear[0] -> linear[1] -> classifier -> activation -> OSP weights in one
    launch) against the three launches it replaces (two k_gemm_pre.hip GEMMs + seg_head_kernel): the
    same arithmetic statement by statement, so the outputs must be IDENTICAL; and against an f64
    restatement of the MLP on the 22-bit operands."""
    g = torch.Generator().manual_seed(17 + B)
    rows, K, classes = B * F, 3, (7 if powerset else 3)
    h = torch.tanh(torch.randn(rows, 256, generator=g))
    W0, b0 = torch.randn(128, 256, generator=g) / 16, torch.randn(128, generator=g) * 0.1
    W1, b1 = torch.randn(128, 128, generator=g) / 11, torch.randn(128, generator=g) * 0.1
    cw = torch.zeros(64, 128)
    cw[:classes] = torch.randn(classes, 128, generator=g) / 8
    cb = torch.zeros(64)
    cb[:classes] = torch.randn(classes, generator=g) * 0.2
    dh, dW0, dW1 = _kb(h).to(gpu), _kb(W0).to(gpu), _kb(W1).to(gpu)
    db0, db1, dcw, dcb = b0.to(gpu), b1.to(gpu), cw.to(gpu), cb.to(gpu)
    lib, ctx = _lib.load(), _ctx(gpu)
    # --- three launches
    m0 = torch.zeros(2, rows, 128, dtype=torch.int16, device=gpu)
    m1 = torch.zeros(rows, 128, device=gpu)
    d = _lib.ConvGemmDesc()
    d.Xsplit, d.xplane, d.Wsplit, d.bias = dh.data_ptr(), rows * 256, dW0.data_ptr(), db0.data_ptr()
    d.Ysplit, d.yplane = m0.data_ptr(), rows * 128
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, rows, rows, rows, 256, 1, 1
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = 256, 256, 128, 128, 256, 128, _lib.EPI_BIAS_LEAKY
    _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d), None), "lin0")
    d2 = _lib.ConvGemmDesc()
    d2.Xsplit, d2.xplane, d2.Wsplit, d2.bias, d2.Y = m0.data_ptr(), rows * 128, dW1.data_ptr(), db1.data_ptr(), m1.data_ptr()
    d2.B, d2.Tin, d2.Tout, d2.Tstore, d2.Cin, d2.taps, d2.dil = 1, rows, rows, rows, 128, 1, 1
    d2.K, d2.Kpad, d2.Npad, d2.Nstore, d2.ldx, d2.ldy, d2.epi = 128, 128, 128, 128, 128, 128, _lib.EPI_BIAS_LEAKY
    _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d2), None), "lin1")
    seg3, w3 = torch.empty(B, F, K, device=gpu), torch.empty(B, K, F, device=gpu)
    _lib.check(lib.dz_k_seg_head(ctx, m1.data_ptr(), dcw.data_ptr(), dcb.data_ptr(), B, F, classes, K, int(powerset),
                                 seg3.data_ptr(), 3.0, 10.0, 0, w3.data_ptr(), None), "seg_head")
    # --- one launch
    seg1 = torch.full((B, F, K), float("nan"), device=gpu)
    w1 = torch.full((B, K, F), float("nan"), device=gpu)
    _lib.check(lib.dz_k_mlp_head(ctx, dh.data_ptr(), rows * 256, dW0.data_ptr(), dW1.data_ptr(), db0.data_ptr(),
                                 db1.data_ptr(), dcw.data_ptr(), dcb.data_ptr(), rows, F, classes, K, int(powerset),
                                 3.0, 10.0, seg1.data_ptr(), w1.data_ptr(), None), "mlp_head")
    _sync()
    assert torch.equal(seg1, seg3) and torch.equal(w1, w3)
    # without the OSP output
    seg0 = torch.empty(B, F, K, device=gpu)
    _lib.check(lib.dz_k_mlp_head(ctx, dh.data_ptr(), rows * 256, dW0.data_ptr(), dW1.data_ptr(), db0.data_ptr(),
                                 db1.data_ptr(), dcw.data_ptr(), dcb.data_ptr(), rows, F, classes, K, int(powerset),
                                 3.0, 10.0, seg0.data_ptr(), None, None), "mlp_head")
    _sync()
    assert torch.equal(seg0, seg3)
    # --- f64 restatement of the MLP (sigmoid case: a smooth function of the logits)
    if not powerset:
        a0 = F_leaky(_unplanes(_planes(h)) @ _unplanes(_planes(W0)).t() + b0.double())
        a1 = F_leaky(a0 @ _unplanes(_planes(W1)).t() + b1.double())
        ref = torch.sigmoid(a1 @ cw[:classes].double().t() + cb[:classes].double())
        assert (seg1.cpu().double().view(rows, K) - ref).abs().max().item() < 2e-6


def F_leaky(x):
    return F.leaky_relu(x, 0.01)


def test_gemm_split_plane_output(gpu):
    """k_gemm_split.hip (f32 input, norm-on-load, per-chunk batches: the tdnn1 call) writing its
    output as f16 planes for a k_gemm_pre.hip consumer."""
    g = torch.Generator().manual_seed(3)
    B, T, Cin, N, taps = 3, 293, 64, 512, 5
    Tout = T - 4
    X = torch.randn(B, T, Cin, generator=g)
    W = torch.randn(N, taps * Cin, generator=g) / math.sqrt(taps * Cin)
    bias, e0, e1 = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    sc, sh = torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.2
    keep = [t.to(gpu) for t in (X, W, _planes(W), bias, e0, e1, sc, sh)]
    Yp = torch.zeros((2, B, T, N), dtype=torch.int16, device=gpu)
    Yf = torch.zeros((B, T, N), device=gpu)
    lib = _lib.load()
    for planes in (False, True):
        d = _lib.ConvGemmDesc()
        d.X, d.W, d.Wsplit, d.bias, d.e0, d.e1, d.nscale, d.nshift = [t.data_ptr() for t in keep]
        if planes:
            d.Ysplit, d.yplane = Yp.data_ptr(), B * T * N
        else:
            d.Y = Yf.data_ptr()
        d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = B, T, Tout, Tout, Cin, taps, 1
        d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.nld = taps * Cin, taps * Cin, N, N, Cin, N, Cin
        d.xbs, d.ybs, d.norm_on_load, d.epi = T * Cin, T * N, 1, _lib.EPI_TDNN
        _lib.check(lib.dz_k_gemm_split(_ctx(gpu), C.byref(d), None), "dz_k_gemm_split")
    _sync()
    want = Yf.cpu()[:, :Tout].double()
    got = _unplanes(_unkb(Yp.cpu(), B * T, N)).view(B, T, N)[:, :Tout]        # kb-major planes of B * T rows
    assert want.abs().max() > 0.5
    assert (got - want).abs().max().item() < 2.0 ** -21 * want.abs().max().item()


@pytest.mark.parametrize("kernel", ["valu", "mfma0", "mfma1", "mfma4"])
def test_lstm_plane_output(gpu, kernel):
    """Both recurrence kernels writing h as f16 (hi, lo) planes == their f32 output split."""
    from diart_amd.weights import lstm_whh_planes
    if kernel == "mfma1" and not _lib.experiments():
        pytest.skip("matrix-core recurrence variant 1 exists in the experiments build only")
    g = torch.Generator().manual_seed(11)
    B, T = 19, 50
    gx = (torch.randn(B, T, 1024, generator=g) * 0.8).to(gpu)
    whh = (torch.rand(2, 512, 128, generator=g) * 2 - 1) * 0.2
    lib = _lib.load()
    hf = torch.zeros(B, T, 256, device=gpu)
    hp = torch.zeros((2, B, T, 256), dtype=torch.int16, device=gpu)
    if kernel == "valu":
        dw = whh.to(gpu)
        _lib.check(lib.dz_k_lstm(_ctx(gpu), gx.data_ptr(), dw.data_ptr(), hf.data_ptr(), B, T, None))
        _lib.check(lib.dz_k_lstm_planes(_ctx(gpu), gx.data_ptr(), dw.data_ptr(), None, 0, hp.data_ptr(),
                                        B * T * 256, B, T, None))
    else:
        v = int(kernel[4])
        dw = lstm_whh_planes(whh, v).to(gpu)
        if v == 4:      # unit-major columns by definition (kernel and dz_k_lstm_planes alike)
            _lib.check(lib.dz_k_lstm_mfma(_ctx(gpu), gx.data_ptr(), dw.data_ptr(), hf.data_ptr(), B, T, 1, v, None))
        else:
            _lib.check(lib.dz_k_lstm_mfma(_ctx(gpu), gx.data_ptr(), dw.data_ptr(), hf.data_ptr(), B, T, 0, v, None))
        _lib.check(lib.dz_k_lstm_planes(_ctx(gpu), gx.data_ptr(), None, dw.data_ptr(), v, hp.data_ptr(),
                                        B * T * 256, B, T, None))
    _sync()
    want = hf.cpu().double()
    got = _unplanes(_unkb(hp.cpu(), B * T, 256)).view(B, T, 256)
    assert want.abs().max() > 0.3
    assert (got - want).abs().max().item() < 2.0 ** -21


@pytest.mark.parametrize("kernel", ["split", "pre"])
def test_split_gemm_dynamic_range(gpu, kernel):
    """The split-f16 arithmetic on inputs an f32 reference handles without thinking: magnitudes from
    1e-7 to 1e4 mixed inside one K row (both kernels keep f32-grade accuracy relative to
    sum |x w|: the scaled low part stays a normal f16 down to |x| = 2^-14 and below that the
    absolute error is < 2^-36)."""
    g = torch.Generator().manual_seed(5)
    M, K, N = 256, 256, 128
    mags = torch.tensor([1e-7, 1e-5, 1e-3, 1.0, 30.0, 1e4])
    X = torch.randn(M, K, generator=g) * mags[torch.randint(0, len(mags), (M, K), generator=g)]
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.zeros(N)
    Y = torch.full((M, N), float("nan"), device=gpu)
    dW, db = (_kb(W) if kernel == "pre" else _planes(W)).to(gpu), bias.to(gpu)
    d = _lib.ConvGemmDesc()
    d.Wsplit, d.bias, d.Y = dW.data_ptr(), db.data_ptr(), Y.data_ptr()
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, M, M, M, K, 1, 1
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = K, K, N, N, K, N, _lib.EPI_BIAS
    if kernel == "pre":
        dX = _kb(X).to(gpu)
        d.Xsplit, d.xplane = dX.data_ptr(), M * K
        _lib.check(_lib.load().dz_k_gemm_pre(_ctx(gpu), C.byref(d), None))
    else:
        dX, dWf = X.to(gpu), W.to(gpu)
        d.X, d.W = dX.data_ptr(), dWf.data_ptr()
        _lib.check(_lib.load().dz_k_gemm_split(_ctx(gpu), C.byref(d), None))
    _sync()
    ref = X.double() @ W.double().t()
    denom = (X.double().abs() @ W.double().abs().t())
    err = ((Y.cpu().double() - ref).abs() / denom).max().item()
    assert err < 1e-6, err        # an f32 GEMM of this depth lands at 1e-7 .. 3e-7 by the same measure


def test_f16x3_dynamic_range_map(gpu, monkeypatch):
    """VERDICT r2 next #6c: where does the split-f16 arithmetic stop being fp32-grade?  Uniform operand
    scales 2^-24 .. 2^15 (activations) and 2^-24 .. 2^8 (weights), one operand at a time, error against
    an f64 reference next to the exact-f32 MFMA kernel's on the same inputs.  The map is printed and
    written to gpurun_out/f16x3_range.json (DESIGN.md 4.4 quotes it).  Asserted: inside
    2^-12 <= scale <= 2^12 (magnitudes ~2^-13 .. 2^14 for N(0,1) data) f16x3 is within 3x of the f32
    kernel's error; below, the ABSOLUTE error floor 2^-36 |w| shows as a relative error that grows with
    1 / scale (never silent garbage)."""
    import json
    import warnings
    from pathlib import Path
    monkeypatch.setenv("DZ_ENGINE", "split_strict=0")  # the map goes below what split_f16 accepts for a real layer
    warnings.simplefilter("ignore")
    g = torch.Generator().manual_seed(11)
    M, K, N = 256, 512, 128
    X0 = torch.randn(M, K, generator=g)
    W0 = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.zeros(N, device=gpu)
    lib = _lib.load()
    rows = []

    def run(X, W):
        ref = X.double() @ W.double().t()
        out = {}
        for kernel in ("f16x3", "f32"):
            Y = torch.full((M, N), float("nan"), device=gpu)
            d = _lib.ConvGemmDesc()
            d.bias, d.Y = bias.data_ptr(), Y.data_ptr()
            d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, M, M, M, K, 1, 1
            d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = K, K, N, N, K, N, _lib.EPI_BIAS
            keep = []
            if kernel == "f16x3":
                dW, dX = _kb(W).to(gpu), _kb(X).to(gpu)
                d.Wsplit, d.Xsplit, d.xplane = dW.data_ptr(), dX.data_ptr(), M * K
                keep += [dW, dX]
                _lib.check(lib.dz_k_gemm_pre(_ctx(gpu), C.byref(d), None))
            else:
                dX, dWf = X.to(gpu), W.to(gpu)
                d.X, d.W = dX.data_ptr(), dWf.data_ptr()
                keep += [dX, dWf]
                _lib.check(lib.dz_k_convgemm(_ctx(gpu), C.byref(d), None))
            _sync()
            out[kernel] = ((Y.cpu().double() - ref).norm() / ref.norm()).item()
        return out

    for which, exps in (("activations", range(-24, 16, 2)), ("weights", range(-24, 10, 2))):
        for e in exps:
            sc = 2.0 ** e
            X, W = (X0 * sc, W0) if which == "activations" else (X0, W0 * sc)
            if X.abs().max() > 65504 or W.abs().max() > 65504:
                # beyond the f16 range: a weight is refused at pack time (weights.split_f16), an activation is
                # flagged by the kernel that writes its planes (test_f16x3_operands_beyond_the_f16_range_...)
                rows.append({"operand": which, "log2_scale": e, "beyond_f16_range": True})
                continue
            r = run(X, W)
            rows.append({"operand": which, "log2_scale": e, "f16x3_rel_l2": r["f16x3"], "f32_rel_l2": r["f32"]})
            print(f"{which:12s} scale 2^{e:<4d} f16x3 {r['f16x3']:.2e}   exact f32 {r['f32']:.2e}")
            if -12 <= e <= 12:
                assert r["f16x3"] <= max(3 * r["f32"], 1e-6), (which, e, r)
            assert r["f16x3"] < 2.0 ** (-35 - min(e, -12)) * 64, (which, e, r)     # graceful: ~2^-36 absolute floor
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "f16x3_range.json").write_text(json.dumps(rows, indent=1))


@pytest.mark.parametrize("kernel", ["f32", "f16x3"])
@pytest.mark.parametrize("taps,dil,Cin,N,T", [(3, 2, 128, 128, 77), (3, 4, 128, 128, 301), (5, 1, 80, 1024, 150)])
def test_same_convolution_with_reflect_padding_and_second_input(gpu, kernel, taps, dil, Cin, N, T):
    """ECAPA-TDNN's "same" convolutions (reflect padding, speechbrain Conv1d) with the Res2Net second input
    (conv(x_i + y_{i-1})) -> ReLU -> folded BatchNorm, on the exact-f32 kernel and (round 3) on the
    split-f16 kernel, against torch's own reflect-padded conv1d in f64."""
    g = torch.Generator().manual_seed(taps * 100 + dil)
    B, pad = 3, (taps - 1) * dil // 2
    x = torch.randn(B, T, Cin, generator=g)
    x2 = torch.randn(B, T, Cin, generator=g) if Cin == 128 else None
    w = torch.randn(N, Cin, taps, generator=g) / math.sqrt(Cin * taps)
    bias, e0, e1 = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    kpad = (taps * Cin + 31) // 32 * 32
    Wp = _pack(w, Cin, N, kpad)
    from diart_amd.weights import split_f16
    keep = [t.to(gpu) for t in (x, Wp, split_f16(Wp), bias, e0, e1)]
    Y = torch.full((B, T, N), float("nan"), device=gpu)
    d = _lib.ConvGemmDesc()
    d.X, d.W, d.bias, d.e0, d.e1, d.Y = keep[0].data_ptr(), keep[1].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr(), Y.data_ptr()
    if x2 is not None:
        dx2 = x2.to(gpu)
        d.X2 = dx2.data_ptr()
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil, d.pad = B, T, T, T, Cin, taps, dil, pad
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy = taps * Cin, kpad, N, N, Cin, N
    d.xbs, d.ybs, d.epi = T * Cin, T * N, _lib.EPI_RELU_BN
    lib = _lib.load()
    if kernel == "f16x3":
        d.Wsplit = keep[2].data_ptr()
        _lib.check(lib.dz_k_gemm_split(_ctx(gpu), C.byref(d), None), "dz_k_gemm_split")
    else:
        _lib.check(lib.dz_k_convgemm(_ctx(gpu), C.byref(d), None), "dz_k_convgemm")
    _sync()
    xin = (x if x2 is None else x + x2).double().permute(0, 2, 1)
    ref = F.conv1d(F.pad(xin, (pad, pad), mode="reflect"), w.double(), bias.double(), dilation=dil)
    ref = (F.relu(ref) * e0.double()[None, :, None] + e1.double()[None, :, None]).permute(0, 2, 1)
    got = Y.cpu().double()
    assert not torch.isnan(got).any()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
