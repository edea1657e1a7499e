#!/bin/bash
# visit AB: where the 0.80 ms of a batch-1 segmentation forward goes (kernel durations vs gaps between dependent launches)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
rm -f gpurun_out/b1_timeline.txt
DZ_PROF_TIMELINE=gpurun_out/b1_timeline.txt timeout -s KILL 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import time, torch, numpy as np
from diart_amd import models as M, _lib
from diart_amd.synth import synth_segmentation_state
lib = _lib.load()
dev = torch.device("cuda", 0)
seg = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=1)
x = torch.randn(1, 1, 80000, device=dev) * 0.1
for _ in range(10): seg(x)
torch.cuda.synchronize()
ts = []
for _ in range(50):
    t0 = time.monotonic(); seg(x); torch.cuda.synchronize(); ts.append(1e3 * (time.monotonic() - t0))
print("wall p50 ms (no brackets)", round(float(np.percentile(ts, 50)), 3))
lib.dz_prof_enable(1)
for _ in range(5):
    seg(x); torch.cuda.synchronize()
    lib.dz_prof_collect()
lib.dz_prof_enable(0)
rows, cur = [], []
for ln in open("gpurun_out/b1_timeline.txt"):
    if ln.startswith("#"):
        if cur: rows.append(cur)
        cur = []
    else:
        t, c, s, d = ln.split(); cur.append((t, float(s), float(d)))
if cur: rows.append(cur)
r = rows[-1]
span = max(s + d for _, s, d in r) - min(s for _, s, d in r)
busy = sum(d for _, s, d in r)
print("launches", len(r), "span us", round(span, 1), "sum of durations us", round(busy, 1), "gaps us", round(span - busy, 1))
prev_end = None
for t, s, d in sorted(r, key=lambda v: v[1]):
    gap = (s - prev_end) if prev_end is not None else 0.0
    print(f"  {t:14s} start {s:8.1f} dur {d:7.1f} gap_before {gap:6.1f}")
    prev_end = max(prev_end or 0.0, s + d)
PY
