#!/usr/bin/env python
"""Where a wave of sinc_conv0_h spends a tile: shader-clock stamps per wave (experiments build) at the loop top, after
the MFMA phase, after parking the next tile's samples, after the result stores and after the tile barrier — and which
SIMD of which CU each wave ran on (HW_ID), i.e. how two resident 3-wave workgroups really land on four SIMDs.
usage: python tools/conv0_phases.py"""
import collections
import json
import os
import sys
from pathlib import Path

os.environ.setdefault("DZ_EXPERIMENTS", "1")
import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib  # noqa: E402
from diart_amd.weights import split_f16  # noqa: E402

dev = torch.device("cuda", 0)
lib, ctx = _lib.load(), _lib.context(0)
_lib.set_option("pack_cache", 1)      # fixed weights: the kernel-level entries pack their operand once
B, S = 64, 80000
st = torch.cuda.current_stream(dev).cuda_stream
nt = lib.dz_k_conv0_split_ntile(S)
wave = torch.randn(B, S, device=dev) * 0.1
y0, part = torch.empty(B, 2658, 80, device=dev), torch.empty(B, nt, 80, 2, device=dev)
stats = torch.zeros(B, 2, device=dev)
stats[:, 1] = 1.0
fs = split_f16(torch.randn(96, 256) * 0.05).to(dev)
run = lambda: _lib.check(lib.dz_k_sinc_conv0_split(ctx, wave.data_ptr(), S, B, S, stats.data_ptr(), 1.0, 0.0, fs.data_ptr(),
                                                   y0.data_ptr(), part.data_ptr(), st))
for _ in range(3):
    run()
NW = 3 if os.environ.get("DZ_CONV0_V2") == "0" else 4          # waves per workgroup: sinc_conv0_h / sinc_conv0_v2
stamps = torch.zeros(512 * NW * 64, dtype=torch.int64, device=dev)
lib.dz_k_conv_pool_debug(stamps.data_ptr())
run()
torch.cuda.synchronize()
lib.dz_k_conv_pool_debug(None)
s = stamps.cpu().numpy().reshape(512, NW, 64)
NAMES = ["fetch issue + MFMA phase (3 blocks x 16 k-steps)", "park next tile", "stores + partials", "wait at the tile barrier"]
hw = s[:, :, 63]
simd, cu, se, sh, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, (hw >> 12) & 1, 0
# waves per (physical CU, SIMD): key CU by (se, sh, cu) — per XCD the ids repeat, so count the distribution of SIMD loads
per_cu = collections.defaultdict(lambda: [0, 0, 0, 0])
for wg in range(512):
    for w in range(NW):
        per_cu[(int(se[wg, w]), int(sh[wg, w]), int(cu[wg, w]))][int(simd[wg, w])] += 1
loads = collections.Counter(tuple(sorted(v, reverse=True)) for v in per_cu.values())
print("waves per SIMD, per (se, sh, cu) id (ids repeat across the 8 XCDs: divide by 8):", dict(loads))
# the SIMDs of the three waves of one workgroup
wg_pat = collections.Counter(tuple(int(x) for x in simd[wg]) for wg in range(512))
print("SIMD of the waves of a workgroup:", dict(wg_pat.most_common(8)))
role = (hw >> 32) & 1                      # v2: 1 = heavy wave
rows = {0: [], 1: [], 2: [], 3: []}
for wg in range(512):
    for w in range(NW):
        v = s[wg, w, :60]
        n = int((v != 0).sum()) // 5
        for t in range(1, n):            # skip the first tile (prologue effects)
            seg = v[5 * t:5 * t + 5].astype(np.float64)
            nxt = v[5 * (t + 1)] if t + 1 < n else None
            rows[w if NW == 3 else int(role[wg, w])].append(np.diff(seg))
out = {}
for w in (range(3) if NW == 3 else range(2)):
    r = np.array(rows[w])
    print(f"{('wave %d' % w) if NW == 3 else ('light waves', 'heavy waves')[w]}: {len(r)} tiles, mean cycles per tile {r.sum(1).mean():.0f}")
    for nm, m, p10, p90 in zip(NAMES, r.mean(0), np.percentile(r, 10, axis=0), np.percentile(r, 90, axis=0)):
        print(f"    {nm:52s} {m:8.0f}   (p10 {p10:.0f}, p90 {p90:.0f})")
    out[f"wave{w}"] = dict(zip(NAMES, [round(float(x)) for x in r.mean(0)]))
if NW == 4 and (s[:, :, 62] != 0).any():
    # the launch as a whole: entry (slot 62) / exit (slot 61) stamps of every wave, first loop stamp in slot 0
    ent, ext, first = s[:, :, 62].astype(np.float64), s[:, :, 61].astype(np.float64), s[:, :, 0].astype(np.float64)
    ok = (ent > 0) & (ext > 0)
    t0 = ent[ok].min()
    ntiles = np.array([(int((s[wg, w, :60] != 0).sum()) // 5) for wg in range(512) for w in range(NW)]).reshape(512, NW)
    print(f"launch, shader clock: first entry -> last exit {ext[ok].max() - t0:.0f} cycles; entry skew (p50 / p90 / max) "
          f"{np.percentile(ent[ok] - t0, 50):.0f} / {np.percentile(ent[ok] - t0, 90):.0f} / {(ent[ok] - t0).max():.0f}; "
          f"prologue = entry -> first tile (mean / p90) {(first[ok] - ent[ok]).mean():.0f} / {np.percentile(first[ok] - ent[ok], 90):.0f}; "
          f"a wave's life entry -> exit (mean) {(ext[ok] - ent[ok]).mean():.0f}; tiles per workgroup {ntiles[:, 0].min()} - {ntiles[:, 0].max()}")
    if (s[:, :, 60] != 0).any():
        st = [s[:, :, k].astype(np.float64) for k in (62, 60, 59, 58, 57, 0)]
        names = ["entry -> roles known (HW_ID publish + barrier)", "-> own prologue loads issued (bank to registers / first tile by LDS-DMA)",
                 "-> chunk statistics (lanes 0 - 1 of wave 0)", "-> statistics barrier", "-> first tile parked + barrier (loop starts)"]
        for r_, lab in ((1, "heavy"), (0, "light")):
            m = ok & (role == r_)
            print(f"  prologue of the {lab} waves, mean cycles: " + "; ".join(f"{n_} {(b_[m] - a_[m]).mean():.0f}" for n_, a_, b_ in zip(names, st, st[1:])))
    out["launch"] = {"first_entry_to_last_exit": float(ext[ok].max() - t0), "entry_skew_p90": float(np.percentile(ent[ok] - t0, 90)),
                     "prologue_mean": float((first[ok] - ent[ok]).mean()), "wave_life_mean": float((ext[ok] - ent[ok]).mean())}
# MFMA phase by how many waves share the SIMD (from HW_ID): co-resident waves = waves with the same (se, sh, cu, simd) ... ids repeat
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/conv0_phases.json").write_text(json.dumps({"phases": out, "simd_loads": {str(k): v for k, v in loads.items()},
                                                            "wg_simd_patterns": {str(k): v for k, v in wg_pat.items()}}, indent=1))
