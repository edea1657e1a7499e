#!/usr/bin/env python
"""How much of the recurrence's in-pipeline slow-down (172 us alone, 270 - 300 us in the 64-stream pipeline) is
EXECUTION under contention and how much is its workgroups waiting for a CU?  One recurrence launch (128 workgroups)
timed on its own stream while GEMM launches run on another: started before the GEMMs (its workgroups are resident when
the contention begins) and after them (they have to find CUs first).
usage: python tools/rec_contention.py [--out gpurun_out/rec_contention.json]"""
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
ap.add_argument("--out", default="gpurun_out/rec_contention.json")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib, ctx = _lib.load(), _lib.context(0)
B, F = 64, 293
gx = torch.randn(B, F, 1024, device=dev) * 0.5
whh = torch.randn(2, 512, 128, device=dev) * 0.1
hout = torch.empty(B, F, 256, device=dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def gemm_desc(M, Cin, N, taps, dil):
    K = taps * Cin
    X, W = torch.randn(M, Cin) * 0.7, torch.randn(N, K) / K ** 0.5
    keep = [kb_major(split_f16(X)).to(dev), kb_major(split_f16(W)).to(dev), torch.zeros(N, device=dev), torch.ones(N, device=dev),
            torch.zeros(2, M * N, dtype=torch.int16, device=dev)]
    d = _lib.ConvGemmDesc()
    d.Xsplit, d.xplane, d.Wsplit = keep[0].data_ptr(), M * Cin, keep[1].data_ptr()
    d.bias, d.e0, d.e1 = keep[2].data_ptr(), keep[3].data_ptr(), keep[2].data_ptr()
    Tout = M - (taps - 1) * dil
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, M, Tout, Tout, Cin, taps, dil
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = K, K, N, N, Cin, N, _lib.EPI_TDNN
    d.Ysplit, d.yplane = keep[4].data_ptr(), M * N
    return d, keep


d, keep = gemm_desc(64 * 289, 512, 512, 3, 2)


def rec():
    _lib.check(lib.dz_k_lstm(ctx, gx.data_ptr(), whh.data_ptr(), hout.data_ptr(), B, F, sA.cuda_stream))


def gemms(gen, n=4):
    # gen 1: k_gemm_pre.hip; 2: k_gemm_g2.hip; >= 100: a timing-only instantiation of generation 2 (DBG = gen - 100:
    # 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads, 8 no barrier — tools/g2ablate.py), to see WHAT the recurrence
    # suffers from when it shares its CU
    for _ in range(n):
        if gen == 1:
            _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d), sB.cuda_stream))
        elif gen == 2:
            _lib.check(lib.dz_k_gemm_g2(ctx, C.byref(d), 2, sB.cuda_stream))
        else:
            _lib.check(lib.dz_k_gemm_g2(ctx, C.byref(d), 2 + 16 * (gen - 100), sB.cuda_stream))


def measure(order, gen):
    tot = 0.0
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if order == "gemm_first":
            gemms(gen)
        e0.record(sA)
        rec()
        e1.record(sA)
        if order == "rec_first":
            gemms(gen)
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return round(tot / args.reps, 1)


for _ in range(3):
    rec()
    gemms(1, 1)
    gemms(2, 1)
torch.cuda.synchronize()
res = {"rec_alone_us": measure("alone", 1)}
for gen in (1, 2):
    for order in ("rec_first", "gemm_first"):
        res[f"rec_us_{order}_gemm_g{gen}"] = measure(order, gen)
NAMES = {1: "no LDS-DMA (MFMA + fragment reads)", 5: "MFMA only", 13: "MFMA only, no barrier", 6: "LDS-DMA only (no MFMA, no reads)",
         3: "fragment reads only", 7: "barriers only", 2: "LDS-DMA + reads, no MFMA"}
for dbg, nm in NAMES.items():
    gemms(100 + dbg, 1)
    torch.cuda.synchronize()
    res["rec_us_beside_g2: " + nm] = measure("rec_first", 100 + dbg)
try:
    _lib.check(lib.dz_range_check(ctx, 1), "range")
except Exception:
    pass
print(json.dumps(res), flush=True)
Path(args.out).parent.mkdir(exist_ok=True)
Path(args.out).write_text(json.dumps(res, indent=1))
