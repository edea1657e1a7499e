#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace --stats result (rocpd .db or *_kernel_stats.csv) into the
small per-kernel summary that is committed under profiles/.
usage: tools/prof_summary.py <prof_dir> <out.md> [title]"""
import csv, glob, sqlite3, sys
from pathlib import Path

src, out = Path(sys.argv[1]), Path(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else src.name
rows = []
dbs = sorted(glob.glob(str(src / "**/*.db"), recursive=True))
csvs = sorted(glob.glob(str(src / "**/*kernel_stats.csv"), recursive=True))
if csvs:
    for r in csv.DictReader(open(csvs[0])):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
elif dbs:
    c = sqlite3.connect(dbs[0])
    for name, calls, tot, avg, pct in c.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        rows.append((name, calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if tot > 1e6 else avg, pct))
else:
    sys.exit(f"no rocprofv3 stats under {src}")
# the db view reports ns in 'duration' (start/end are ns); normalise to microseconds
if dbs and not csvs:
    c = sqlite3.connect(dbs[0])
    rows = []
    for name, calls, tot, mn, mx in c.execute(
            "select name,count(*),sum(duration),min(duration),max(duration) from kernels group by name order by sum(duration) desc"):
        rows.append((name, calls, tot / 1e3, tot / calls / 1e3, 0.0))
    total = sum(r[2] for r in rows)
    rows = [(n, c_, t, a, 100.0 * t / total) for n, c_, t, a, _ in rows]
# medians from the raw trace (robust to a warm-up launch that sat behind an allocation)
med = {}
traces = sorted(glob.glob(str(src / "**/*kernel_trace.csv"), recursive=True))
if traces:
    import statistics
    per = {}
    for r in csv.DictReader(open(traces[0])):
        per.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    med = {k: statistics.median(v) for k, v in per.items()}
with open(out, "w") as f:
    f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats (durations in microseconds; median from the raw "
            f"kernel trace of the same run)\n\n")
    f.write("| kernel | calls | total us | avg us | median us | % |\n|---|---:|---:|---:|---:|---:|\n")
    for n, c_, t, a, p in rows:
        m = f"{med[n]:.2f}" if n in med else ""
        f.write(f"| `{n[:110]}` | {c_} | {t:.1f} | {a:.2f} | {m} | {p:.2f} |\n")
print(open(out).read())
