#!/usr/bin/env python3
"""Is the 21 - 39 ms start-up stall of the traced runs (profiles/README.md) there without a tracer?
A fresh process, the bench's 64-stream pipeline, wall time between consecutive finish() returns of
the first steps; prints the per-step times and the largest one.  Run several processes:

  for i in 1 2 3 4 5 6 7 8; do python tools/stall_probe.py; done"""
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from diart_amd.models import HipEmbedding, HipSegmentation  # noqa: E402
from diart_amd.pipeline import StreamBatch  # noqa: E402
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_streams  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    n, hop, S = 64, 8000, 80000
    audio = torch.from_numpy(synth_streams(n, (S + hop * (steps + 1)) / 16000.0, seed0=0)).to(dev)
    pipe = StreamBatch(HipSegmentation(synth_segmentation_state(), max_batch=n, precision="f16x3"),
                       HipEmbedding(synth_embedding_state(), max_batch=n, precision="f16x3"), n, device=dev,
                       cluster_threads=8, tail=True)
    torch.cuda.synchronize()
    warm = os.environ.get("DZ_PROBE_WARM", "")
    if warm:
        t0 = time.perf_counter()
        pin = torch.empty(1024, dtype=torch.float32).pin_memory()
        devbuf = torch.zeros(1024, device=dev)
        for lane in pipe.lanes:
            for s in list(lane["a"]) + list(lane["b"]):
                with torch.cuda.stream(s):
                    if "k" in warm:
                        devbuf.add_(1.0)                      # a kernel on this stream
                    if "c" in warm:
                        pin.copy_(devbuf, non_blocking=True)  # a D2H copy to pinned memory on this stream
                    if "e" in warm:
                        ev = torch.cuda.Event()
                        ev.record(s)
                        ev.synchronize()
        torch.cuda.synchronize()
        print("warm", warm, round((time.perf_counter() - t0) * 1e3, 1), "ms", file=sys.stderr)
    hw = os.environ.get("DZ_PROBE_HOSTWARM")     # which part of the FIRST host half freezes the queues?
    if hw is not None:                           # GPU-only warm steps (pretend a real step has happened) ...
        import numpy as np
        pipe._real_steps = 1
        F, K, D = pipe.seg.num_frames(S), pipe.seg.num_speakers, pipe.emb.dimension
        if "clu" in hw:                          # ... plus, with the GPU idle, the first clustering call (worker threads)
            pipe.clustering(np.zeros((n, F, K), np.float32), np.zeros((n, K, D), np.float32), True)
            pipe.clustering.reset()
        if "tail" in hw:                         # ... plus the output tail's buffers and its first call
            from diart_amd.blocks.aggregation import BatchedOutputTail
            pipe.tail = BatchedOutputTail(n, F, pipe.max_speakers, pipe.step, pipe.latency, pipe.tau_active,
                                          num_threads=pipe.cluster_threads)
            pipe.tail(np.zeros((n, F, pipe.max_speakers)), np.zeros(n), pipe.duration / F)
            pipe.tail.reset()
    if os.environ.get("DZ_PROBE_FULLWARM"):      # complete steps (GPU + host half) on silence, then reset the streams
        z = torch.zeros((n, S), device=dev)
        for _ in range(int(os.environ["DZ_PROBE_FULLWARM"])):
            pipe.finish(pipe.launch(z))
        pipe.reset()
        torch.cuda.synchronize()
    tl = os.environ.get("DZ_PROF_TIMELINE")
    if tl:                                   # every launch of every step bracketed with its own timestamps
        from diart_amd import _lib
        Path(tl).unlink(missing_ok=True)
        _lib.load().dz_prof_enable(1)
    inflight, stamps = [], []
    for t in range(steps):
        inflight.append(pipe.launch(audio[:, t * hop:t * hop + S]))
        if len(inflight) >= pipe.max_inflight:
            pipe.finish(inflight.pop(0))
            stamps.append(time.perf_counter())
    while inflight:
        pipe.finish(inflight.pop(0))
        stamps.append(time.perf_counter())
    if tl:
        _lib.load().dz_prof_collect()
        rows = [l.split() for l in open(tl) if not l.startswith("#")]
        rows = [(r[0], float(r[2]), float(r[3])) for r in rows]
        long_ = [(n, round(t0 / 1e3, 2), round(d / 1e3, 2)) for n, t0, d in rows if d > 3000]
        ends = sorted((t0, t0 + d) for _, t0, d in rows)
        gaps, cur = [], ends[0][1]
        for a, b in ends[1:]:
            if a - cur > 3000:
                gaps.append((round(cur / 1e3, 2), round((a - cur) / 1e3, 2)))
            cur = max(cur, b)
        print(json.dumps({"launches": len(rows), "longer_than_3ms (tag, start_ms, dur_ms)": long_,
                          "idle_gaps_over_3ms (at_ms, dur_ms)": gaps}))
    d = [(b - a) * 1e3 for a, b in zip(stamps, stamps[1:])]
    worst = max(range(len(d)), key=lambda i: d[i])
    print(json.dumps({"per_step_ms_first_12": [round(x, 2) for x in d[:12]], "median_ms": round(sorted(d)[len(d) // 2], 3),
                      "max_ms": round(d[worst], 2), "max_at_step": worst + 1,
                      "over_5ms (step, ms)": [(i + 1, round(x, 1)) for i, x in enumerate(d) if x > 5.0]}))


if __name__ == "__main__":
    main()
