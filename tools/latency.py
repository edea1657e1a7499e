#!/usr/bin/env python
"""BASELINE.json config 5: VoiceActivityDetection pipeline, step 250 ms, one stream, batch 1 —
per-chunk latency of ``pipeline([chunk])`` measured like diart's Chronometer
(/root/reference/src/diart/utils.py:13-43: time.monotonic around the call, H2D/D2H included).
Also the full SpeakerDiarization pipeline at batch 1 and the bare segmentation forward.
Reference points (README.md:169-171): segmentation 12 ms CPU / 8 ms GPU (RTX 4060),
+ embedding 26 ms CPU / 12 ms GPU."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import models as M  # noqa: E402
from diart_amd.blocks import (SpeakerDiarization, SpeakerDiarizationConfig, VoiceActivityDetection,  # noqa: E402
                              VoiceActivityDetectionConfig)
from diart_amd.features import SlidingWindow, SlidingWindowFeature  # noqa: E402
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream  # noqa: E402

from diart_amd.hostinfo import limit_host_threads  # noqa: E402

print("host threads:", torch.get_num_threads(), "->", limit_host_threads(), file=sys.stderr)
dev = torch.device("cuda", 0)
SR = 16000
n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 400


def chunks(stream, step):
    S, H = 5 * SR, int(round(step * SR))
    for i in range((len(stream) - S) // H + 1):
        yield SlidingWindowFeature(stream[i * H:i * H + S, None], SlidingWindow(start=i * step, duration=1 / SR, step=1 / SR))


def measure(pipe, step, n):
    stream = synth_stream(5, 5.0 + step * (n + 8))
    times = []
    for i, c in enumerate(chunks(stream, step)):
        t0 = time.monotonic()
        pipe([c])
        times.append(1e3 * (time.monotonic() - t0))
        if i >= n + 5:
            break
    t = np.array(times[5:])
    return {"p50_ms": round(float(np.percentile(t, 50)), 3), "p95_ms": round(float(np.percentile(t, 95)), 3),
            "mean_ms": round(float(t.mean()), 3), "chunks": int(t.size)}


seg = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=1)
emb = M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=1)
out = {}
vad = VoiceActivityDetection(VoiceActivityDetectionConfig(segmentation=seg, step=0.25, device=dev))
out["vad_pipeline_step250ms_b1"] = measure(vad, 0.25, n_chunks)
dia = SpeakerDiarization(SpeakerDiarizationConfig(segmentation=seg, embedding=emb, device=dev))
out["diarization_pipeline_step500ms_b1"] = measure(dia, 0.5, n_chunks // 2)
# bare forward, input resident on the GPU
x = torch.randn(1, 1, 80000, device=dev) * 0.1
for name, fn in (("segmentation_forward_b1", lambda: seg(x)), ("embedding_forward_3rows", lambda: emb(x.repeat(3, 1, 1), torch.rand(3, 293, device=dev)))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(100):
        t0 = time.monotonic()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.monotonic() - t0))
    out[name] = {"p50_ms": round(float(np.percentile(ts, 50)), 3), "p95_ms": round(float(np.percentile(ts, 95)), 3)}
print(json.dumps(out, indent=1))
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/latency.json").write_text(json.dumps(out, indent=1))
