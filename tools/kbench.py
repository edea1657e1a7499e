#!/usr/bin/env python
"""Isolated timing of the HIP kernels at the config-2 shapes (64 chunks), one kernel at a time
on an otherwise idle GPU (torch events on the stream the kernel is launched on).
usage: python tools/kbench.py [--batch 64] [--only lstm,tdnn2,...]"""
import argparse
import os
import ctypes as C
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--only", type=str, default="")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--agroup", type=int, default=0)
ap.add_argument("--power", type=float, default=0.0,
                help="seconds each kernel is looped for while a side thread samples the card's hwmon files (shader clock, "
                     "package power): adds clock / watts / joules per launch to every row")
args = ap.parse_args()
only = set(filter(None, args.only.split(",")))
dev = torch.device("cuda", 0)
lib = _lib.load()
_lib.set_option("pack_cache", 1)      # fixed weights: the kernel-level entries pack their register-resident operand once
ctx = _lib.context(0)
B = args.batch
st = torch.cuda.current_stream(dev).cuda_stream
results = {}


from diart_amd.hwmon import PowerSampler  # noqa: E402

sampler = PowerSampler(device_index=0) if args.power > 0 else None
power_windows = []        # (name, t0, t1, launches)


def timeit(name, fn, flop=None, bytes_=None):
    if only and name not in only:
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    row = {"us": round(us, 1)}
    if flop:
        row["tflops"] = round(flop / us / 1e6, 2)
    if bytes_:
        row["gbps"] = round(bytes_ / us / 1e3, 1)
    results[name] = row
    print(f"{name:14s} {us:9.1f} us  {row.get('tflops', '')} TF  {row.get('gbps', '')} GB/s", flush=True)
    if sampler is not None:
        import time
        n = max(args.reps, int(args.power * 1e6 / max(us, 1.0)))
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        t1 = time.time()
        power_windows.append((name, t0 + 0.15 * (t1 - t0), t1 - 0.02, n, (t1 - t0) / n))


def convgemm(name, Bn, Tin, Cin, N, taps, dil, epi, Npad=None, pro=False, pool=False, ksplit=0):
    Npad = Npad or N
    Tout = Tin - (taps - 1) * dil
    K = taps * Cin
    Kpad = (K + 31) // 32 * 32
    X = torch.randn(Bn, Tin, Cin, device=dev)
    W = torch.randn(Npad, Kpad, device=dev) * 0.05
    bias = torch.zeros(Npad, device=dev)
    e0 = torch.ones(Npad, device=dev)
    Tstore = Tout // 3 if pool else Tout
    Y = torch.empty(max(ksplit, 1) * Bn, Tstore, Npad, device=dev)
    d = _lib.ConvGemmDesc()
    d.X, d.W, d.bias, d.Y, d.e0, d.e1 = (X.data_ptr(), W.data_ptr(), bias.data_ptr(), Y.data_ptr(),
                                         e0.data_ptr(), bias.data_ptr())
    keep = [X, W, bias, e0, Y]
    if pro:
        sc = torch.ones(Bn, Cin, device=dev)
        d.nscale, d.nshift, d.nld, d.norm_on_load = sc.data_ptr(), sc.data_ptr(), Cin, 1
        keep.append(sc)
    if pool:
        part = torch.empty(Bn, lib.dz_k_convgemm_ntile(Tout), Npad, 2, device=dev)
        d.partials = part.data_ptr()
        keep.append(part)
    d.B, d.Tin, d.Tout, d.Cin, d.taps, d.dil = Bn, Tin, Tout, Cin, taps, dil
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.Tstore = K, Kpad, Npad, Npad, Cin, Npad, Tstore
    d.xbs, d.ybs, d.epi = Tin * Cin, Tstore * Npad, epi
    d.ksplit, d.ysplit = ksplit, Bn * Tstore * Npad
    d.agroup = args.agroup
    flop = 2.0 * Bn * Tout * K * N
    _lib.set_option("f32_gemm", 0)                        # the round-1 exact-f32 kernel (k_convgemm.hip)
    timeit(name, lambda: _lib.check(lib.dz_k_convgemm(ctx, C.byref(d), st), name), flop=flop)
    _lib.set_option("f32_gemm", 1)
    if not pro and not pool and not ksplit and Npad % 128 == 0 and Cin % 32 == 0 and (not only or name in only or name + "_f32g2" in only):
        # k_gemm_f32.hip: what dz_k_convgemm routes these layers to by default
        y_old = Y.clone()
        Y.zero_()
        only_saved = set(only)
        only.clear()
        timeit(name + "_f32g2", lambda: _lib.check(lib.dz_k_gemm_f32(ctx, C.byref(d), st), name), flop=flop)
        only.update(only_saved)
        torch.cuda.synchronize()
        results[name + "_f32g2"]["max_abs_vs_round1_kernel"] = float((Y - y_old).abs().max().item())
    if not ksplit and (Npad % 128 == 0 or pool) and (not only or name + "_split" in only or name in only):
        from diart_amd.weights import split_f16
        ws = split_f16(W.cpu()).to(dev)
        y32 = Y.clone()
        Y.zero_()
        d.Wsplit = ws.data_ptr()
        keep.append(ws)
        only_saved = set(only)
        only.clear()
        timeit(name + "_split", lambda: _lib.check(lib.dz_k_gemm_split(ctx, C.byref(d), st), name + "_split"), flop=flop)
        only.update(only_saved)
        torch.cuda.synchronize()
        err = ((Y - y32).norm() / y32.norm()).item()
        mx = ((Y - y32).abs().max() / y32.abs().max()).item()
        results[name + "_split"]["rel_l2_vs_f32"] = err
        print(f"    {name}_split vs f32 kernel: rel L2 {err:.2e}, max|d|/max|y| {mx:.2e}", flush=True)
        if pool:
            Y.zero_()
            only.clear()
            timeit(name + "_convpool", lambda: _lib.check(lib.dz_k_conv_pool(ctx, C.byref(d), st), name), flop=flop)
            only.update(only_saved)
            torch.cuda.synchronize()
            err = ((Y - y32).norm() / y32.norm()).item()
            results[name + "_convpool"]["rel_l2_vs_f32"] = err
            print(f"    {name}_convpool vs f32 kernel: rel L2 {err:.2e}", flush=True)
    if not ksplit and not pro and not pool and Cin % 32 == 0 and Npad % 128 == 0 and (not only or name + "_pre" in only or name in only):
        # k_gemm_pre.hip: flattened rows, both operands as f16 planes, f32 and plane output
        from diart_amd.weights import kb_major, split_f16
        M = Bn * Tin
        xs = kb_major(split_f16(X.reshape(M, Cin).cpu())).to(dev)
        ws = kb_major(split_f16(W.cpu())).to(dev)
        Yf = torch.zeros(M, Npad, device=dev)
        Yp = torch.zeros(2, M, Npad, dtype=torch.int16, device=dev)
        d2 = _lib.ConvGemmDesc()
        d2.Xsplit, d2.xplane, d2.Wsplit, d2.bias, d2.e0, d2.e1 = xs.data_ptr(), M * Cin, ws.data_ptr(), bias.data_ptr(), e0.data_ptr(), bias.data_ptr()
        d2.B, d2.Tin, d2.Tout, d2.Tstore, d2.Cin, d2.taps, d2.dil = 1, M, M - (taps - 1) * dil, M - (taps - 1) * dil, Cin, taps, dil
        d2.K, d2.Kpad, d2.Npad, d2.Nstore, d2.ldx, d2.ldy, d2.epi = K, K, Npad, Npad, Cin, Npad, epi
        d2.agroup = args.agroup
        only_saved = set(only)
        only.clear()
        d2.Y = Yf.data_ptr()
        timeit(name + "_pre_f32out", lambda: _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d2), st), name), flop=flop)
        d2.Y, d2.Ysplit, d2.yplane = None, Yp.data_ptr(), M * Npad
        timeit(name + "_pre", lambda: _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d2), st), name), flop=flop)
        only.update(only_saved)
        torch.cuda.synchronize()
        # compare the rows every chunk computes validly with the f32 kernel's result
        y32 = Y.reshape(Bn, Tstore, Npad)
        got = Yf.reshape(Bn, Tin, Npad)[:, :Tstore]
        err = ((got - y32).norm() / y32.norm()).item()
        results[name + "_pre"]["rel_l2_vs_f32"] = err
        print(f"    {name}_pre vs f32 kernel: rel L2 {err:.2e}", flush=True)
        keep += [xs, ws, Yf, Yp]
    return keep


F = 293
# ---- LSTM recurrence --------------------------------------------------------------------
gx = torch.randn(B, F, 1024, device=dev) * 0.5
whh = torch.randn(2, 512, 128, device=dev) * 0.1
hout = torch.empty(B, F, 256, device=dev)
timeit("lstm", lambda: _lib.check(lib.dz_k_lstm(ctx, gx.data_ptr(), whh.data_ptr(), hout.data_ptr(), B, F, st)),
       flop=2.0 * B * F * 2 * 512 * 128)
from diart_amd.weights import lstm_whh_planes  # noqa: E402
hout2 = torch.empty(B, F, 256, device=dev)
for variant in ((0, 1, 2, 3, 4) if _lib.experiments() else (0, 3, 4)):      # 1 / 2: experiments build only
    whs = lstm_whh_planes(whh.cpu(), variant).to(dev)
    nm = f"lstm_mfma{variant}"
    gxv = gx.view(B, F, 2, 4, 128).transpose(3, 4).reshape(B, F, 1024).contiguous() if variant >= 3 else gx
    if variant >= 4:      # (its x-projection carries the gates' activation scales)
        from diart_amd.weights import LSTM_GATE_SCALE
        gxv = (gxv.view(B, F, 256, 4) * torch.tensor(LSTM_GATE_SCALE, device=dev)).view(B, F, 1024).contiguous()
    timeit(nm, lambda: _lib.check(lib.dz_k_lstm_mfma(ctx, gxv.data_ptr(), whs.data_ptr(), hout2.data_ptr(), B, F, int(variant >= 3), variant, st)),
           flop=2.0 * B * F * 2 * 512 * 128)
    if "lstm" in results and nm in results:
        torch.cuda.synchronize()
        d = (hout2 - hout).abs().max().item()
        results[nm]["max_abs_vs_valu"] = d
        print(f"    {nm} vs lstm (f32 VALU): max|d| {d:.2e}", flush=True)
# ---- sinc conv0 -----------------------------------------------------------------------------
wave = torch.randn(B, 80000, device=dev) * 0.1
stats = torch.zeros(B, 2, device=dev)
stats[:, 1] = 1.0
filt = torch.randn(128, 96, device=dev) * 0.05
y0 = torch.empty(B, 2658, 80, device=dev)
part0 = torch.empty(B, 42, 80, 2, device=dev)
timeit("wave_stats", lambda: _lib.check(lib.dz_k_wave_stats(ctx, wave.data_ptr(), 80000, B, 80000, stats.data_ptr(), st)),
       bytes_=B * 80000 * 4.0)
timeit("sinc_conv0", lambda: _lib.check(lib.dz_k_sinc_conv0(ctx, wave.data_ptr(), 80000, B, 80000, stats.data_ptr(), 1.0, 0.0,
                                                           filt.data_ptr(), y0.data_ptr(), part0.data_ptr(), st)),
       flop=2.0 * B * 7975 * 251 * 80)
from diart_amd.weights import split_f16 as _sp16  # noqa: E402
fsplit = _sp16(torch.randn(96, 256) * 0.05).to(dev)
nt_s = lib.dz_k_conv0_split_ntile(80000)
part0s = torch.empty(B, nt_s, 80, 2, device=dev)
y0s = torch.empty(B, 2658, 80, device=dev)
timeit("sinc_conv0_split", lambda: _lib.check(lib.dz_k_sinc_conv0_split(ctx, wave.data_ptr(), 80000, B, 80000, stats.data_ptr(), 1.0, 0.0,
                                                                       fsplit.data_ptr(), y0s.data_ptr(), part0s.data_ptr(), st)),
       flop=2.0 * B * 7975 * 251 * 80)
# both networks' first stage in one launch (round 5; experiments build: DZ_EXPERIMENTS=1)
if _lib.experiments():
    pair_planes = _sp16(torch.randn(192, 256) * 0.05).to(dev)
    pair_bsum = torch.zeros(192, device=dev)
    mom = torch.empty(B, lib.dz_wave_stats_floats(), device=dev)
    _lib.check(lib.dz_wave_stats(ctx, wave.data_ptr(), 80000, B, 80000, mom.data_ptr(), st))
    y0e, part0e = torch.empty(B, 2658, 80, device=dev), torch.empty(B, nt_s, 80, 2, device=dev)
    timeit("sinc_conv0_pair", lambda: _lib.check(lib.dz_k_sinc_conv0_pair(ctx, wave.data_ptr(), 80000, B, 80000, mom.data_ptr(),
                                                                         pair_planes.data_ptr(), pair_bsum.data_ptr(), 1.0, 1.0,
                                                                         y0s.data_ptr(), y0e.data_ptr(), part0s.data_ptr(),
                                                                         part0e.data_ptr(), st)),
           flop=2.0 * B * 7975 * 251 * 160)
# ---- implicit-GEMM layers ------------------------------------------------------------------
convgemm("conv1_pool", B, 2658, 80, 60, 5, 1, _lib.EPI_POOL3, Npad=64, pro=True, pool=True)
convgemm("conv2_pool", B, 884, 64, 60, 5, 1, _lib.EPI_POOL3, Npad=64, pro=True, pool=True)
convgemm("lstm_proj0", B, F, 64, 1024, 1, 1, _lib.EPI_BIAS, pro=True)
convgemm("lstm_proj", 1, B * F, 256, 1024, 1, 1, _lib.EPI_BIAS)
convgemm("seg_mlp0", 1, B * F, 256, 128, 1, 1, _lib.EPI_BIAS_LEAKY)
convgemm("tdnn1", B, 293, 64, 512, 5, 1, _lib.EPI_TDNN, pro=True)
convgemm("tdnn2", B, 289, 512, 512, 3, 2, _lib.EPI_TDNN)
convgemm("tdnn3", B, 285, 512, 512, 3, 3, _lib.EPI_TDNN)
convgemm("tdnn4", B, 279, 512, 512, 1, 1, _lib.EPI_TDNN)
convgemm("tdnn5", B, 279, 512, 1500, 1, 1, _lib.EPI_TDNN, Npad=1536)
# the pipeline runs tdnn2..5 FLATTENED over all B * P rows (api.hip emb_frames): the form the exact-f32 kernels see
convgemm("tdnn2_flat", 1, B * 293, 512, 512, 3, 2, _lib.EPI_TDNN)
convgemm("tdnn5_flat", 1, B * 293, 512, 1500, 1, 1, _lib.EPI_TDNN, Npad=1536)
convgemm("emb_linear", 1, B * 3, 3008, 512, 1, 1, _lib.EPI_BIAS, ksplit=16)
# ---- stats pooling ----------------------------------------------------------------------------
x5 = torch.randn(B, 279, 1536, device=dev)
w = torch.rand(B * 3, F, device=dev)
pooled = torch.empty(B * 3, 3008, device=dev)
timeit("stats_pool", lambda: _lib.check(lib.dz_k_stats_pool(ctx, x5.data_ptr(), 279, 1500, 1536, w.data_ptr(), F, B * 3, 3,
                                                           pooled.data_ptr(), 3008, st)),
       bytes_=B * 279 * 1536 * 4.0)
# ---- segmentation tail: two MLP GEMMs + head (three launches) vs k_mlp_head.hip (one) --------
if not only or "mlp_head" in only:
    from diart_amd.weights import kb_major, split_f16
    rows = B * F
    hs = kb_major(split_f16(torch.tanh(torch.randn(rows, 256)))).to(dev)
    w0s, w1s = kb_major(split_f16(torch.randn(128, 256) / 16)).to(dev), kb_major(split_f16(torch.randn(128, 128) / 11)).to(dev)
    b0, b1 = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    cw, cb = torch.randn(64, 128, device=dev) / 8, torch.zeros(64, device=dev)
    segb, wb = torch.empty(B, F, 3, device=dev), torch.empty(B, 3, F, device=dev)
    m1 = torch.randn(rows, 128, device=dev)
    only_saved = set(only)
    only.clear()
    timeit("mlp_head", lambda: _lib.check(lib.dz_k_mlp_head(ctx, hs.data_ptr(), rows * 256, w0s.data_ptr(), w1s.data_ptr(),
                                                            b0.data_ptr(), b1.data_ptr(), cw.data_ptr(), cb.data_ptr(), rows, F, 3, 3, 0,
                                                            3.0, 10.0, segb.data_ptr(), wb.data_ptr(), st)),
           flop=2.0 * rows * (256 * 128 + 128 * 128 + 128 * 3))
    timeit("seg_head", lambda: _lib.check(lib.dz_k_seg_head(ctx, m1.data_ptr(), cw.data_ptr(), cb.data_ptr(), B, F, 3, 3, 0,
                                                            segb.data_ptr(), 3.0, 10.0, 0, wb.data_ptr(), st)),
           bytes_=rows * 134 * 4.0)
    only.update(only_saved)
if sampler is not None:
    import time
    time.sleep(1.5)
    t_idle = time.time()
    time.sleep(0.5)
    sampler.stop()
    card = sampler.card()
    _, idle_w, _ = sampler.window(t_idle, t_idle + 0.5, card)
    print(f"power: card {card} ({'by PCI address' if sampler.own is not None else 'by power span'}), idle {idle_w:.0f} W")
    for name, t0, t1, n, sec in power_windows:
        mhz, watts, _ = sampler.window(t0, t1, card)
        if watts is None:
            continue
        results[name].update({"loop_us": round(sec * 1e6, 1), "sclk_mhz": round(mhz), "package_w": round(watts),
                              "joules_per_launch": round(watts * sec, 4), "joules_above_idle": round((watts - idle_w) * sec, 4)})
        print(f"{name:22s} looped {sec * 1e6:8.1f} us  {mhz:5.0f} MHz  {watts:5.0f} W  {watts * sec * 1e3:7.2f} mJ / launch")
out = Path("gpurun_out")
out.mkdir(exist_ok=True)
(out / "kbench.json").write_text(json.dumps(results, indent=1))
