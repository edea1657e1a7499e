#!/usr/bin/env python
"""Generation 1 (k_gemm_pre.hip) against generation 2 (k_gemm_g2.hip) of the pre-split GEMM at the config-2
shapes: results (against an f64 product of the SAME split operands' f32 values and against each other) and
time per launch — alone on an idle GPU, and while LSTM recurrence workgroups hold 128 / 256 of the CUs (the
regime half the chip is in inside the 64-stream pipeline: one wave per SIMD beside two recurrence waves).

usage: python tools/g2bench.py [--reps 20] [--only tdnn2,proj] [--out gpurun_out/g2bench.json]"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import os
os.environ.setdefault("DZ_EXPERIMENTS", "1")     # needs libdiart_amd_exp.so (python -m diart_amd.build --experiments)
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib  # noqa: E402
from diart_amd.weights import kb_major, split_f16  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", type=str, default="")
ap.add_argument("--out", type=str, default="gpurun_out/g2bench.json")
ap.add_argument("--no-rec", action="store_true")
args = ap.parse_args()
only = set(filter(None, args.only.split(",")))
dev = torch.device("cuda", 0)
lib = _lib.load()
ctx = _lib.context(0)
B = args.batch
F = 293
results = {}
s_main = torch.cuda.Stream(dev)
s_rec = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

# LSTM recurrence used as the co-resident load: 2 B chains = 128 workgroups of 512 threads, 168 VGPRs
gx = torch.randn(B, F, 1024, device=dev) * 0.5
whh = torch.randn(2, 512, 128, device=dev) * 0.1
hout = [torch.empty(B, F, 256, device=dev) for _ in range(2)]


def rec(i):
    _lib.check(lib.dz_k_lstm(ctx, gx.data_ptr(), whh.data_ptr(), hout[i].data_ptr(), B, F, s_rec[i].cuda_stream))


def timeit(fn, nrec=0):
    """mean us per launch of fn on s_main; nrec recurrence launches (one per side stream) are started right
    before every launch, and everything is drained between repetitions"""
    st = s_main.cuda_stream
    for _ in range(3):
        fn(st)
    torch.cuda.synchronize()
    tot = 0.0
    if nrec == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s_main)
        for _ in range(args.reps):
            fn(st)
        e1.record(s_main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(nrec):
            rec(i)
        e0.record(s_main)
        fn(st)
        e1.record(s_main)
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / args.reps


def layer(name, rows_in, Cin, N, taps, dil, epi, plane_out):
    if only and name not in only:
        return
    K = taps * Cin
    M = rows_in
    Tout = M - (taps - 1) * dil
    X = torch.randn(M, Cin) * 0.7
    W = torch.randn(N, K) * (1.0 / K ** 0.5)
    bias = (torch.randn(N) * 0.1).to(dev)
    e0 = (torch.rand(N) + 0.5).to(dev)
    e1 = (torch.randn(N) * 0.1).to(dev)
    xs = kb_major(split_f16(X)).to(dev)
    ws = kb_major(split_f16(W)).to(dev)
    Yf = torch.zeros(M, N, device=dev)
    Yp = torch.zeros(2, M * N, dtype=torch.int16, device=dev)
    d = _lib.ConvGemmDesc()
    d.Xsplit, d.xplane, d.Wsplit = xs.data_ptr(), M * Cin, ws.data_ptr()
    d.bias, d.e0, d.e1 = bias.data_ptr(), e0.data_ptr(), e1.data_ptr()
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, M, Tout, Tout, Cin, taps, dil
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = K, K, N, N, Cin, N, epi
    flop = 2.0 * Tout * K * N

    def set_out(plane):
        if plane:
            d.Y, d.Ysplit, d.yplane = None, Yp.data_ptr(), M * N
        else:
            d.Y, d.Ysplit, d.yplane = Yf.data_ptr(), None, 0

    def run(gen, mt):
        if gen == 1:
            return lambda st: _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d), st), name)
        if gen == 3:
            return lambda st: _lib.check(lib.dz_k_gemm_g3(ctx, C.byref(d), mt, st), name)
        return lambda st: _lib.check(lib.dz_k_gemm_g2(ctx, C.byref(d), mt, st), name)

    # ---- results: f32 output of every variant against the f64 product of the operands' f32 values ----
    Xd, Wd = X.double().to(dev), W.double().to(dev)
    acc = torch.zeros(Tout, N, dtype=torch.float64, device=dev)
    for tp in range(taps):
        acc += Xd[tp * dil: tp * dil + Tout] @ Wd[:, tp * Cin:(tp + 1) * Cin].T
    acc += bias.double()
    if epi == _lib.EPI_BIAS_LEAKY:
        acc = torch.where(acc > 0, acc, acc * 0.01)
    elif epi == _lib.EPI_TDNN:
        acc = torch.where(acc > 0, acc, acc * 0.01) * e0.double() + e1.double()
    row = {"shape": [M, K, N], "gflop": round(flop / 1e9, 2)}
    set_out(False)
    outs = {}
    ALL = (("g1", 1, 0), ("g2_mt2", 2, 2), ("g2_mt3", 2, 3), ("g2_mt4", 2, 4), ("g3_mt2", 3, 2), ("g3_mt3", 3, 3), ("g3_mt4", 3, 4))
    for tag, gen, mt in ALL:
        with torch.cuda.stream(s_main):
            Yf.zero_()
            run(gen, mt)(s_main.cuda_stream)
        torch.cuda.synchronize()
        got = Yf[:Tout].double()
        outs[tag] = Yf[:Tout].clone()
        row[tag + "_rel_l2_vs_f64"] = float(((got - acc).norm() / acc.norm()).item())
        row[tag + "_max_abs_vs_f64"] = float((got - acc).abs().max().item())
    for tag in [t for t, _, _ in ALL[1:]]:
        row[tag + "_max_abs_vs_g1"] = float((outs[tag] - outs["g1"]).abs().max().item())
    # plane output: gen 2 planes against gen 1 planes (hi must agree except at rounding ties of different sums)
    set_out(True)
    planes = {}
    for tag, gen, mt in (("g1", 1, 0), ("g2_mt2", 2, 2), ("g3_mt4", 3, 4)):
        with torch.cuda.stream(s_main):
            Yp.zero_()
            run(gen, mt)(s_main.cuda_stream)
        torch.cuda.synchronize()
        v = Yp.view(torch.float16).float()
        planes[tag] = v[0] + v[1] / 2048.0
    for tag in ("g2_mt2", "g3_mt4"):
        row[tag + "_planes_max_abs_vs_g1"] = float((planes[tag] - planes["g1"]).abs().max().item())
    _lib.check(lib.dz_range_check(ctx, 1), "range")
    # ---- time ----
    set_out(plane_out)
    for tag, gen, mt in ALL:
        for nrec in ((0,) if args.no_rec else (0, 1, 2)):
            us = timeit(run(gen, mt), nrec)
            row[f"{tag}_us_rec{nrec}"] = round(us, 1)
    row["ideal_us_f16x3"] = round(3 * flop / 2.5e15 * 1e6, 1)
    results[name] = row
    print(name, json.dumps(row), flush=True)


layer("tdnn2", B * 289, 512, 512, 3, 2, _lib.EPI_TDNN, True)
layer("tdnn3", B * 285, 512, 512, 3, 3, _lib.EPI_TDNN, True)
layer("tdnn4", B * 279, 512, 512, 1, 1, _lib.EPI_TDNN, True)
layer("tdnn5", B * 279, 512, 1536, 1, 1, _lib.EPI_TDNN, False)
layer("proj", B * F, 256, 1024, 1, 1, _lib.EPI_BIAS, False)
layer("mlp0", B * F, 256, 128, 1, 1, _lib.EPI_BIAS_LEAKY, True)
if not args.no_rec:
    us = timeit(lambda st: _lib.check(lib.dz_k_lstm(ctx, gx.data_ptr(), whh.data_ptr(), hout[0].data_ptr(), B, F, st)), 0)
    results["lstm_rec_alone_us"] = round(us, 1)
    print("lstm_rec alone", round(us, 1), flush=True)
out = Path(args.out)
out.parent.mkdir(exist_ok=True)
out.write_text(json.dumps(results, indent=1))
