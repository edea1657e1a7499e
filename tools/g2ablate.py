#!/usr/bin/env python
"""Where a k-tile of k_gemm_g2.hip spends its time: the DBG instantiations (no LDS-DMA / no MFMA / no fragment
reads / no barrier inside the loop; results are wrong) on a grid of exactly one 128 x 128 tile per CU, and
k_gemm_pre.hip on a grid of exactly two per CU.  K = 1536 (48 k-tiles of 24 MFMAs per wave).
usage: python tools/g2ablate.py [--out gpurun_out/g2ablate.json]"""
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
ap.add_argument("--out", default="gpurun_out/g2ablate.json")
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib, ctx = _lib.load(), _lib.context(0)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
res = {"cus": ncu}


def desc(M, Cin, N, taps, dil, plane, fill="random"):
    K = taps * Cin
    X, W = torch.randn(M, Cin) * 0.7, torch.randn(N, K) / K ** 0.5
    if fill == "zeros":
        X, W = torch.zeros(M, Cin), torch.zeros(N, K)
    elif fill == "const":        # hi planes constant, lo planes zero
        X, W = torch.full((M, Cin), 0.5), torch.full((N, K), 0.25)
    keep = [kb_major(split_f16(X)).to(dev), kb_major(split_f16(W)).to(dev), torch.zeros(N, device=dev), torch.ones(N, device=dev),
            torch.zeros(M, N, device=dev), torch.zeros(2, M * N, dtype=torch.int16, device=dev)]
    d = _lib.ConvGemmDesc()
    d.Xsplit, d.xplane, d.Wsplit = keep[0].data_ptr(), M * Cin, keep[1].data_ptr()
    d.bias, d.e0, d.e1 = keep[2].data_ptr(), keep[3].data_ptr(), keep[2].data_ptr()
    Tout = M - (taps - 1) * dil
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, M, Tout, Tout, Cin, taps, dil
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = K, K, N, N, Cin, N, _lib.EPI_TDNN
    if plane:
        d.Ysplit, d.yplane = keep[5].data_ptr(), M * N
    else:
        d.Y = keep[4].data_ptr()
    return d, keep


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / args.reps, 1)


NAMES = {0: "as built", 1: "no DMA", 2: "no MFMA", 3: "reads + barrier only", 5: "MFMA + barrier (no DMA, no reads)",
         6: "DMA + barrier (no MFMA, no reads)", 7: "barrier only", 13: "MFMA only", 15: "empty loop"}
for label, M in (("one tile per CU", 128 * (ncu // 4)), ("tdnn2 (2.2 tiles per CU)", 64 * 289)):
    d, keep = desc(M + 4, 512, 512, 3, 2, True)       # Tout = M
    row = {}
    for dbg, nm in NAMES.items():
        mt = 2 + 16 * dbg
        row[nm] = timeit(lambda: _lib.check(lib.dz_k_gemm_g2(ctx, C.byref(d), mt, None), nm))
    d2, keep2 = desc(2 * M + 4 if label.startswith("one") else M + 4, 512, 512, 3, 2, True)
    row["k_gemm_pre.hip" + (" (two tiles per CU)" if label.startswith("one") else "")] = timeit(
        lambda: _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d2), None), "g1"))
    res[label] = row
    print(label, json.dumps(row), flush=True)
    try:
        _lib.check(lib.dz_range_check(ctx, 1), "range")
    except Exception:
        pass
ideal = 48 * 24 * 32 / 2.4e3
res["ideal_us_per_tile_at_2.4GHz"] = round(ideal, 1)
Path(args.out).parent.mkdir(exist_ok=True)
Path(args.out).write_text(json.dumps(res, indent=1))

# ---- generation 3 on shapes without / with shared tiles (256 x 128 tiles) ----------------------------------------
for label, M in (("g3: one 256x128 tile per CU (no sharing)", 256 * (ncu // 4)), ("g3: tdnn2", 64 * 289),
                 ("g3: 1.5 tiles per CU", 384 * (ncu // 4))):
    d, keep = desc(M + 4, 512, 512, 3, 2, True)
    row = {}
    for nm, fn in (("g2 mt4", lambda: lib.dz_k_gemm_g2(ctx, C.byref(d), 4, None)),
                   ("g3 mt4", lambda: lib.dz_k_gemm_g3(ctx, C.byref(d), 4, None)),
                   ("g3 mt2", lambda: lib.dz_k_gemm_g3(ctx, C.byref(d), 2, None)),
                   ("g1", lambda: lib.dz_k_gemm_pre(ctx, C.byref(d), None))):
        row[nm] = timeit(lambda: _lib.check(fn(), nm))
    res[label] = row
    print(label, json.dumps(row), flush=True)
Path(args.out).write_text(json.dumps(res, indent=1))

# ---- is the ~40 % wall the power budget?  the same launches on zero / constant operands (DVFS give-back) ----------
M = 128 * (ncu // 4) * 2
for fill in ("random", "const", "zeros"):
    d, keep = desc(M + 4, 512, 512, 3, 2, True, fill)
    row = {}
    for nm, fn in (("g1", lambda: lib.dz_k_gemm_pre(ctx, C.byref(d), None)),
                   ("g2 mt4", lambda: lib.dz_k_gemm_g2(ctx, C.byref(d), 4, None)),
                   ("g3 mt4", lambda: lib.dz_k_gemm_g3(ctx, C.byref(d), 4, None))):
        row[nm] = timeit(lambda: _lib.check(fn(), nm))
    res["fill " + fill] = row
    print("two 128x128 tiles per CU, operands", fill, json.dumps(row), flush=True)
Path(args.out).write_text(json.dumps(res, indent=1))
