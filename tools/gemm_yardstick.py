#!/usr/bin/env python
"""An independent yardstick for the "1.0 PF wall" of the split-f16 GEMMs.

The vendor's f16 / bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS; tools/ only, never the product) at the
shapes of the x-vector TDNN layers and the LSTM projection, on random and on zero operands, next to
k_gemm_pre.hip (`dz_k_gemm_pre`) on the same box in the same process:

  * "f16 K"   — the plain f16 product at the layer's own K: what ONE MFMA per algorithmic product costs;
  * "f16 3K"  — the same M x N with 3 K: the MFMA work the split-f16 ("f16x3") kernel issues for the layer
                (three f16 MFMAs per product), i.e. the like-for-like comparison of matrix-pipe throughput;
  * "ours"    — k_gemm_pre.hip, f32 in / out precision through two f16 planes per operand.

If the library also stops near 1.0 PFLOP/s on random operands (and speeds up on zeros) the wall is the chip's;
if it reaches >= 1.4 PF there is schedule headroom in k_gemm_pre.hip.

usage: python tools/gemm_yardstick.py [--reps 30] [--out gpurun_out/gemm_yardstick.json]"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib  # noqa: E402
from diart_amd.weights import kb_major, split_f16  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--out", type=str, default="gpurun_out/gemm_yardstick.json")
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib, ctx = _lib.load(), _lib.context(0)
results = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "reps": args.reps, "layers": {}}
try:
    results["blas_library"] = str(torch.backends.cuda.preferred_blas_library())
except Exception as exc:  # noqa: BLE001
    results["blas_library"] = repr(exc)


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.reps       # us per call


def vendor(M, N, K, dtype, zero):
    a = (torch.zeros if zero else torch.randn)(M, K, device=dev, dtype=dtype)
    b = (torch.zeros if zero else torch.randn)(N, K, device=dev, dtype=dtype)       # (N, K): the weight layout
    out = torch.empty(M, N, device=dev, dtype=dtype)
    us = timeit(lambda: torch.matmul(a, b.t(), out=out))
    return {"us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}


def ours(M, N, K, zero, plane_out):
    X = torch.zeros(M, K) if zero else torch.randn(M, K) * 0.7
    W = torch.zeros(N, K) if zero else torch.randn(N, K) / K ** 0.5
    xs, ws = kb_major(split_f16(X)).to(dev), kb_major(split_f16(W)).to(dev)
    bias, e0, e1 = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
    Yf = torch.zeros(M, N, device=dev)
    Yp = torch.zeros(2, M * N, dtype=torch.int16, device=dev)
    d = _lib.ConvGemmDesc()
    d.Xsplit, d.xplane, d.Wsplit = xs.data_ptr(), M * K, ws.data_ptr()
    d.bias, d.e0, d.e1 = bias.data_ptr(), e0.data_ptr(), e1.data_ptr()
    d.B, d.Tin, d.Tout, d.Tstore, d.Cin, d.taps, d.dil = 1, M, M, M, K, 1, 1
    d.K, d.Kpad, d.Npad, d.Nstore, d.ldx, d.ldy, d.epi = K, K, N, N, K, N, _lib.EPI_TDNN
    if plane_out:
        d.Y, d.Ysplit, d.yplane = None, Yp.data_ptr(), M * N
    else:
        d.Y, d.Ysplit, d.yplane = Yf.data_ptr(), None, 0
    st = torch.cuda.current_stream().cuda_stream
    us = timeit(lambda: _lib.check(lib.dz_k_gemm_pre(ctx, C.byref(d), st), "dz_k_gemm_pre"))
    alg = 2.0 * M * N * K / us / 1e6
    return {"us": round(us, 1), "alg_tflops": round(alg, 1), "mfma_tflops": round(3 * alg, 1)}


B, F = 64, 293
LAYERS = [("tdnn2 / tdnn3 (K = 3 x 512)", B * F, 512, 1536, True), ("tdnn4", B * F, 512, 512, True),
          ("tdnn5", B * F, 1536, 512, False), ("lstm projection", B * F, 1024, 256, False),
          ("reference point: 8192^3", 8192, 8192, 8192, False)]
for name, M, N, K, plane in LAYERS:
    row = {"M": M, "N": N, "K": K}
    for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for zero in (False, True):
            z = "zeros" if zero else "random"
            row[f"vendor_{tag}_K_{z}"] = vendor(M, N, K, dt, zero)
            if M * N * K * 3 < 2e12:
                row[f"vendor_{tag}_3K_{z}"] = vendor(M, N, 3 * K, dt, zero)
    if M * K < (1 << 28) and N % 128 == 0 and K % 32 == 0 and M * N * K < 1e11:
        for zero in (False, True):
            row[f"ours_f16x3_{'zeros' if zero else 'random'}"] = ours(M, N, K, zero, plane)
    results["layers"][name] = row
    print(name, json.dumps(row), flush=True)
out = Path(args.out)
out.parent.mkdir(parents=True, exist_ok=True)
out.write_text(json.dumps(results, indent=1))
