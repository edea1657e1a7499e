#!/usr/bin/env python
"""Launches that took far longer than their kernel's median, over several rocprofv3 --kernel-trace runs
of the same command (a 21 - 23 ms stats_pool launch in two of four profiled visits).
For every run: the dispatches whose duration exceeds 20x the median of their kernel (and 2 ms), with their
position in the run — a first-launch effect (code-object load, first touch of an arena, a pinned allocation
made while kernels run) shows up at a small dispatch index / start offset, a steady-state stall anywhere.
usage: tools/stall_report.py <out.json> <prof_dir> [<prof_dir> ...]"""
import csv, glob, json, statistics, sys
from pathlib import Path

out, runs = Path(sys.argv[1]), []
for d in sys.argv[2:]:
    traces = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
    if not traces:
        runs.append({"run": d, "error": "no kernel trace"})
        continue
    rows = list(csv.DictReader(open(traces[0])))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    per = {}
    for i, r in enumerate(rows):
        per.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    med = {k: statistics.median(v) for k, v in per.items()}
    seen, slow = {}, []
    for i, r in enumerate(sorted(rows, key=lambda r: int(r["Start_Timestamp"]))):
        k = r["Kernel_Name"]
        seen[k] = seen.get(k, 0) + 1
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if us > 2000.0 and us > 20.0 * med[k]:
            slow.append({"kernel": k[:80], "ms": round(us / 1e3, 2), "median_us": round(med[k], 1), "dispatch_index": i,
                         "nth_launch_of_this_kernel": seen[k], "start_offset_ms": round((int(r["Start_Timestamp"]) - t0) / 1e6, 1)})
    longest = max(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    runs.append({"run": d, "dispatches": len(rows), "stalled": slow,
                 "longest": {"kernel": longest["Kernel_Name"][:80],
                             "ms": round((int(longest["End_Timestamp"]) - int(longest["Start_Timestamp"])) / 1e6, 3)}})
out.write_text(json.dumps({"runs": runs}, indent=1))
for r in runs:
    print(r["run"], "dispatches", r.get("dispatches"), "stalled:", r.get("stalled"), "longest:", r.get("longest"))
