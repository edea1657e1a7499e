#!/usr/bin/env python3
"""Schedule of the 64-stream step from the dispatches' own timestamps (DZ_PROF_TIMELINE, csrc/api.hip
dz_prof_collect): no tracer in the process, so the host runs at its normal speed (a rocprofv3 kernel trace
makes every launch ~4x more expensive on the host and the run host-bound).

  DZ_PROF_EVERY=1 DZ_PROF_TIMELINE=gpurun_out/tl.txt python bench.py --steps 60 ...
  python tools/timeline.py gpurun_out/tl.txt [--steps 40:56] [--json out.json]

Per step (launch order in the file = host enqueue order): when its segmentation chain started / ended, how
much of that span no kernel of the chain was running (gaps), when the embedding frames ran, when the part
behind the segmentation (pooled tdnn5 .. copies) ran; per lane: the time between the end of one step's
segmentation chain and the start of the next one on the same lane."""
import argparse
import json
import sys


def read(path):
    drains, cur = [], None
    for line in open(path):
        if line.startswith("#"):
            cur = []
            drains.append(cur)
            continue
        tag, units, t0, dur = line.split()
        cur.append((tag, int(units), float(t0), float(dur)))
    return drains


def steps_of(rows):
    """Split one drain into steps at `wave_stats`; label the two SincNets by position."""
    steps, cur = [], None
    for r in rows:
        if r[0] == "wave_stats":
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append(r)
    out = []
    for st in steps:
        seg, emb, tail, seen_conv0 = [], [], [], 0
        for tag, units, t0, dur in st[1:]:
            k = dict(tag=tag, t0=t0, t1=t0 + dur, dur=dur)
            if tag == "sinc_conv0":
                seen_conv0 += 1
            if tag in ("tdnn5", "stats_pool", "emb_linear", "l2norm"):
                tail.append(k)
            elif tag.startswith("tdnn") or (seen_conv0 == 2 and tag in ("sinc_conv0", "conv1_pool", "conv2_pool", "finalize_norm")):
                emb.append(k)
            else:
                seg.append(k)
        if seg and emb and tail:
            out.append(dict(stats=st[0][2], seg=seg, emb=emb, tail=tail))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--steps", default="", help="first:last step of the largest drain to print")
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    drains = read(a.file)
    rows = max(drains, key=len)
    st = steps_of(rows)
    lo, hi = (int(x) for x in a.steps.split(":")) if a.steps else (len(st) // 2, min(len(st), len(st) // 2 + 12))
    summary = []
    print(f"{len(st)} steps in the largest drain; steps {lo}..{hi - 1}; times in us relative to the step's wave_stats")
    print(" step  period | seg: start   end   busy   gaps | rec x4 (start:dur) | emb frames start..end | tail start..end | after prev seg on lane")
    for i in range(lo, hi):
        s = st[i]
        z = s["stats"]
        seg0, seg1 = s["seg"][0]["t0"], s["seg"][-1]["t1"]
        busy = sum(k["dur"] for k in s["seg"])
        recs = " ".join(f"{k['t0'] - z:5.0f}:{k['dur']:3.0f}" for k in s["seg"] if k["tag"] == "lstm_rec")
        e0, e1 = s["emb"][0]["t0"], max(k["t1"] for k in s["emb"])
        t0, t1 = s["tail"][0]["t0"], max(k["t1"] for k in s["tail"])
        prev = st[i - a.lanes] if i >= a.lanes else None
        lane_gap = seg0 - prev["seg"][-1]["t1"] if prev else float("nan")
        period = z - st[i - 1]["stats"] if i else float("nan")
        print(f"{i:5d} {period:7.0f} | {seg0 - z:6.0f} {seg1 - z:6.0f} {busy:6.0f} {seg1 - seg0 - busy:6.0f} | {recs} | "
              f"{e0 - z:6.0f}..{e1 - z:6.0f} | {t0 - z:6.0f}..{t1 - z:6.0f} | {lane_gap:7.0f}")
        summary.append(dict(step=i, period_us=period, seg_span_us=seg1 - seg0, seg_busy_us=busy, tail_start_us=t0 - z,
                            tail_end_us=t1 - z, lane_gap_us=lane_gap, emb_frames_us=[e0 - z, e1 - z]))
    n = max(1, hi - lo - 1)
    per = (st[hi - 1]["stats"] - st[lo]["stats"]) / n
    print(f"mean period {per:.0f} us; mean seg span {sum(x['seg_span_us'] for x in summary) / len(summary):.0f} us, "
          f"busy {sum(x['seg_busy_us'] for x in summary) / len(summary):.0f} us; "
          f"mean lane gap {sum(x['lane_gap_us'] for x in summary if x['lane_gap_us'] == x['lane_gap_us']) / len(summary):.0f} us; "
          f"mean step latency {sum(x['tail_end_us'] for x in summary) / len(summary):.0f} us")
    if a.json:
        json.dump(dict(mean_period_us=per, steps=summary), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
