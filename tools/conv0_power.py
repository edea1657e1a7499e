#!/usr/bin/env python
"""Is `sinc_conv0_h` bound by the chip's power budget like the GEMMs?  The same launch (64 chunks) on
random samples and random filters, on zero samples, on zero filters and on both zero: a kernel limited by issue /
latency / LDS takes the same time whatever the values; one limited by power runs faster on zeros."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib  # noqa: E402
from diart_amd.weights import split_f16  # noqa: E402

dev = torch.device("cuda", 0)
lib, ctx = _lib.load(), _lib.context(0)
B, S = 64, 80000
st = torch.cuda.current_stream(dev).cuda_stream
nt = lib.dz_k_conv0_split_ntile(S)
y0, part = torch.empty(B, 2658, 80, device=dev), torch.empty(B, nt, 80, 2, device=dev)
stats = torch.zeros(B, 2, device=dev)
stats[:, 1] = 1.0
out = {}
for wname, wave in (("random samples", torch.randn(B, S, device=dev) * 0.1), ("zero samples", torch.zeros(B, S, device=dev))):
    for fname, filt in (("random filters", torch.randn(96, 256) * 0.05), ("zero filters", torch.zeros(96, 256))):
        fs = split_f16(filt).to(dev)
        fn = lambda: _lib.check(lib.dz_k_sinc_conv0_split(ctx, wave.data_ptr(), S, B, S, stats.data_ptr(), 1.0, 0.0,
                                                          fs.data_ptr(), y0.data_ptr(), part.data_ptr(), st))
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        out[f"{wname}, {fname}"] = round(us, 1)
        print(f"{wname:16s} {fname:16s} {us:7.1f} us", flush=True)
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/conv0_power.json").write_text(json.dumps(out, indent=1))
