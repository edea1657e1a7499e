#!/usr/bin/env python
"""Where a conv_pool_h tile spends its time: shader-clock stamps of the phases (dz_k_conv_pool_debug),
config-2 shape (64 chunks), both layers.  usage: python tools/conv_pool_phases.py
DZ_CONV_POOL_V2=1: the same for conv_pool_v2 (matrix / service waves)."""
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
_lib.set_option("pack_cache", 1)      # fixed weights: the kernel-level entries pack their operand once
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
    if os.environ.get("DZ_CONV_POOL_V2") == "1":      # conv_pool_v2: 8 waves per workgroup, 5 stamps per iteration
        stamps = torch.zeros(256 * 8 * 64, dtype=torch.int64, device=dev)
        lib.dz_k_conv_pool_debug(stamps.data_ptr())
        _lib.check(lib.dz_k_conv_pool(ctx, C.byref(d), st), name)
        torch.cuda.synchronize()
        lib.dz_k_conv_pool_debug(None)
        s = stamps.cpu().numpy().reshape(256, 8, 64)
        place = {}
        for role, label, names in ((1, "matrix waves", ["norm update", "MFMA phase + block hand-over", "-", "-", "tile barrier"]),
                                   (0, "service waves", ["norm update", "exchange reads of tile t-1 + fetch issue of tile t+1",
                                                         "finish tile t-1 (max, stores, sums) + partials of tile t-2",
                                                         "park tile t+1 (wait loads, normalise, split, LDS writes)", "tile barrier"])):
            rows = []
            for wg in range(256):
                sims = tuple(int((s[wg, w, 63] >> 4) & 3) for w in range(8) if s[wg, w, 63])
                if role == 1 and sims:
                    roles = tuple(int((s[wg, w, 63] >> 32) & 1) for w in range(8))
                    place[(sims, roles)] = place.get((sims, roles), 0) + 1
                for w in range(8):
                    v = s[wg, w]
                    if not v[63] or ((int(v[63]) >> 32) & 1) != role:
                        continue
                    n = int((v[:60] != 0).sum())
                    its = n // 5
                    for it in range(its):
                        seg = v[5 * it:5 * it + 5]
                        nxt = v[5 * it + 5] if 5 * it + 5 < n else None
                        dd = list(np.diff(seg))
                        # phase 0 of the NEXT iteration starts at this one's last stamp
                        rows.append((it, its, dd))
            # middle iterations only (steady state): skip the first two and the last two of every wave
            mid = np.array([dd for it, its, dd in rows if 2 <= it < its - 2], dtype=np.float64)
            if len(mid) == 0:
                mid = np.array([dd for it, its, dd in rows], dtype=np.float64)
            print(f"{name} {label}: {len(mid)} steady-state iterations, mean cycles {mid.sum(1).mean():.0f} (from the first stamp to after the barrier)")
            for nm, m, p10, p90 in zip(names[1:] , mid.mean(0), np.percentile(mid, 10, 0), np.percentile(mid, 90, 0)):
                print(f"    {m:8.0f} (p10 {p10:6.0f}, p90 {p90:6.0f})  {nm}")
        print("   (SIMD of waves 0..7, roles) x workgroups:", sorted(place.items(), key=lambda kv: -kv[1])[:4])
        continue
    stamps = torch.zeros(512 * 2 * 64, dtype=torch.int64, device=dev)
    lib.dz_k_conv_pool_debug(stamps.data_ptr())
    _lib.check(lib.dz_k_conv_pool(ctx, C.byref(d), st), name)
    torch.cuda.synchronize()
    lib.dz_k_conv_pool_debug(None)
    s = stamps.cpu().numpy().reshape(512, 2, 64)
    if (s[:, :, 62] != 0).any():
        ok = (s[:, :, 62] != 0) & (s[:, :, 0] != 0)
        pro = (s[:, :, 0] - s[:, :, 62])[ok].astype(np.float64)
        ntl = np.array([int((s[wg, 0, :62] != 0).sum()) // 7 for wg in range(512)])
        print(f"{name}: prologue (kernel entry -> first tile's first stamp) mean {pro.mean():.0f}, p90 {np.percentile(pro, 90):.0f} cycles; "
              f"tiles per workgroup {ntl[ntl > 0].min()} - {ntl.max()}")
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
