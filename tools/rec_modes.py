#!/usr/bin/env python
"""Where a step of the software-pipelined recurrence (k_lstm_mfma.hip, variant 4) goes: the kernel alone on an idle
GPU in its shipped form and — EXPERIMENTS build only (DZ_EXPERIMENTS=1) — in the timing-only forms whose results are
wrong: no stores of h (+4), no wait for the x-projection (+8), no cell arithmetic (+16), combinations.
usage: [DZ_EXPERIMENTS=1] python tools/rec_modes.py [--batch 64] [--frames 293] [--out file.json]"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib  # noqa: E402
from diart_amd.weights import LSTM_GATE_SCALE, lstm_whh_planes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=293)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--out", type=str, default="")
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib = _lib.load()
ctx = _lib.context(0)
B, F = args.batch, args.frames
st = torch.cuda.current_stream(dev).cuda_stream
gx = torch.randn(B, F, 1024, device=dev) * 0.5
whh = torch.randn(2, 512, 128) * 0.1
gx_um = gx.view(B, F, 2, 4, 128).transpose(3, 4).reshape(B, F, 1024).contiguous()
gx_sc = (gx_um.view(B, F, 256, 4) * torch.tensor(LSTM_GATE_SCALE, device=dev)).view(B, F, 1024).contiguous()
hplane = B * F * 256
planes = torch.empty(2 * hplane, dtype=torch.int16, device=dev)
rows = {}
modes = [(3, "variant 3 (round 5)"), (4, "variant 4")]
if _lib.experiments():
    modes += [(8, "4: no stores"), (20, "4: no cell arithmetic"), (24, "4: no stores, no cell arithmetic"),
              (56, "4: no stores, no cells, no LDS-DMA"), (36, "4: no LDS-DMA"), (132, "4: no barriers"),
              (184, "4: MFMAs + LDS reads only")]
for variant, label in modes:
    w = lstm_whh_planes(whh, 4 if variant >= 4 else variant).to(dev)
    g = gx_um if variant == 3 else gx_sc

    def fn():
        _lib.check(lib.dz_k_lstm_planes(ctx, g.data_ptr(), None, w.data_ptr(), variant, planes.data_ptr(), hplane, B, F, st))
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
    rows[label] = {"variant": variant, "us": round(us, 1), "us_per_step": round(us / F, 3), "cycles_per_step_2p4GHz": round(us / F * 2400)}
    print(f"{label:34s} {us:8.1f} us  {us / F:6.3f} us/step  {us / F * 2400:6.0f} cycles/step", flush=True)
if args.out:
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps({"batch": B, "frames": F, "planes_only": True, "rows": rows}, indent=1))
