#!/usr/bin/env python
"""Where a conv_pool_h tile spends its time: shader-clock stamps of the phases (dz_k_conv_pool_debug),
config-2 shape (64 chunks), both layers.  usage: python tools/conv_pool_phases.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import os
os.environ.setdefault("DZ_EXPERIMENTS", "1")     # needs libdiart_amd_exp.so (python -m diart_amd.build --experiments)
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib  # noqa: E402
from diart_amd.weights import split_f16  # noqa: E402

dev = torch.device("cuda", 0)
lib, ctx = _lib.load(), _lib.context(0)
st = torch.cuda.current_stream(dev).cuda_stream
NAMES = ["fetch issue -> barrier A (norm update, wait prev readers)", "park (wait loads, normalise, split, LDS writes)",
         "barrier B", "MFMA phase", "exchange write + barrier C", "epilogue (kh 0) / idle (kh 1)"]
for name, B, Tin, Cin in (("conv1", 64, 2658, 80), ("conv2", 64, 884, 64)):
    Tout, K = Tin - 4, 5 * Cin
    Kpad = (K + 31) // 32 * 32
    X = torch.randn(B, Tin, Cin, device=dev)
    W = torch.randn(64, Kpad) * 0.05
    ws = split_f16(W).to(dev)
    bias = torch.zeros(64, device=dev)
    sc = torch.ones(B, Cin, device=dev)
    Y = torch.empty(B, Tout // 3, 64, device=dev)
    part = torch.empty(B, lib.dz_k_convgemm_ntile(Tout), 64, 2, device=dev)
    d = _lib.ConvGemmDesc()
    d.X, d.W, d.Wsplit, d.bias, d.Y, d.partials = X.data_ptr(), ws.data_ptr(), ws.data_ptr(), bias.data_ptr(), Y.data_ptr(), part.data_ptr()
    d.nscale, d.nshift, d.nld, d.norm_on_load = sc.data_ptr(), sc.data_ptr(), Cin, 1
    d.B, d.Tin, d.Tout, d.Cin, d.taps, d.dil = B, Tin, Tout, Cin, 5, 1
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.Tstore = K, Kpad, 64, 64, Cin, 64, Tout // 3
    d.xbs, d.ybs, d.epi = Tin * Cin, (Tout // 3) * 64, _lib.EPI_POOL3
    for _ in range(3):
        _lib.check(lib.dz_k_conv_pool(ctx, C.byref(d), st), name)
    stamps = torch.zeros(512 * 2 * 64, dtype=torch.int64, device=dev)
    lib.dz_k_conv_pool_debug(stamps.data_ptr())
    _lib.check(lib.dz_k_conv_pool(ctx, C.byref(d), st), name)
    torch.cuda.synchronize()
    lib.dz_k_conv_pool_debug(None)
    s = stamps.cpu().numpy().reshape(512, 2, 64)
    for wv, label in ((0, "wave 0 (k-half 0: epilogue)"), (1, "wave 3 (k-half 1)")):
        rows = []
        for wg in range(512):
            v = s[wg, wv]
            n = int((v != 0).sum())
            for t in range(n // 7):
                seg = v[7 * t:7 * t + 7]
                rows.append(np.diff(seg))
        rows = np.array(rows, dtype=np.float64)
        print(f"{name} {label}: {len(rows)} tiles, mean cycles per phase (total {rows.sum(1).mean():.0f}):")
        for nm, m, p50 in zip(NAMES, rows.mean(0), np.median(rows, 0)):
            print(f"    {m:8.0f} (median {p50:7.0f})  {nm}")
