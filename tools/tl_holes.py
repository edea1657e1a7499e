#!/usr/bin/env python3
"""GPU-side view of a bench pass from the dispatches' own timestamps (DZ_PROF_TIMELINE, see tools/timeline.py):
idle holes and stretched kernels (is a slow period the GPU's or the host's?), and — --bins MS — how many GEMM-shaped
kernels and how many recurrences were running in each interval of every drain (ramp and drain of a 20-step region).

  DZ_PROF_EVERY=1 DZ_PROF_TIMELINE=gpurun_out/tl.txt python bench.py --steps 400 --overlap-brackets ...
  python tools/tl_holes.py gpurun_out/tl.txt [--bins 0.5]"""
import argparse
import collections

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--bins", type=float, default=0.0)
args = ap.parse_args()
drains, cur = [], None
for line in open(args.path):
    if line.startswith("#"):
        cur = []
        drains.append(cur)
        continue
    tag, units, t0, dur = line.split()
    cur.append((float(t0), float(dur), tag))
rows = sorted(r for d in drains for r in d)
print("launches", len(rows), "drains", [len(d) for d in drains], "span ms", round((rows[-1][0] + rows[-1][1] - rows[0][0]) / 1e3, 2))
med = collections.defaultdict(list)
for t0, d, tag in rows:
    med[tag].append(d)
med = {k: sorted(v)[len(v) // 2] for k, v in med.items()}
longk = [(round(t0 / 1e3, 2), tag, round(d / 1e3, 2), round(med[tag] / 1e3, 3)) for t0, d, tag in rows if d > 3000 and d > 4 * med[tag]]
print("kernels > 3 ms and > 4x their median:", len(longk), longk[:20])
end, holes = rows[0][0], []
for t0, d, tag in rows:
    if t0 - end > 500:
        holes.append((round(end / 1e3, 2), round((t0 - end) / 1e3, 2)))
    end = max(end, t0 + d)
print("idle holes > 0.5 ms (at ms, length ms):", holes[:40])
if args.bins > 0:
    w = args.bins * 1e3
    for k, dr in enumerate(drains):
        if len(dr) < 100:
            continue
        dr = sorted(dr)
        T0, T1 = dr[0][0], max(t + d for t, d, _ in dr)
        print(f"drain {k}: {len(dr)} launches, {(T1 - T0) / 1e3:.2f} ms; per {args.bins} ms: (t, GEMM-shaped kernels running, recurrences running)")
        t, line = T0, []
        while t < T1:
            g = r = 0.0
            for t0, d, tag in dr:
                a, b = max(t0, t), min(t0 + d, t + w)
                if b > a:
                    if tag in ("lstm", "lstm_rec", "lstm_mfma"):
                        r += b - a
                    else:
                        g += b - a
            line.append(f"{(t - T0) / 1e3:.1f}:{g / w:.1f}/{r / w:.1f}")
            t += w
        print("  " + "  ".join(line))
