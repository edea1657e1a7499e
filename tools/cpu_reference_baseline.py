#!/usr/bin/env python
"""The CPU baseline SURVEY.md 8(d) describes, in the only place it can run: the build container, where
/root/reference exists.  The REFERENCE'S OWN pipeline class (blocks/diarization.py SpeakerDiarization with its
segmentation / embedding blocks, OnlineSpeakerClustering, DelayedAggregation, Binarize, loaded by path through
oracle/pyannote_stub.py) around the restated networks of oracle/models_ref.py, fed like ``Benchmark`` feeds it
(/root/reference/src/diart/inference.py:275, :392-432): batches of 32 consecutive windows of one file.

Beside it, bench.py's own CPU leg (the oracle's restatement of the same blocks, the one that also runs on the GPU box)
on the same machine, so that the two can be compared: the glue is not what the time goes to.

usage: PYTHONDONTWRITEBYTECODE=1 python tools/cpu_reference_baseline.py [--repeats 50] [--threads 8] [--out profiles/...json]"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ap = argparse.ArgumentParser()
ap.add_argument("--repeats", type=int, default=50)
ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--out", default=str(ROOT / "profiles" / "r04_cpu_reference_baseline_buildbox.json"))
args = ap.parse_args()
torch.set_num_threads(args.threads)

from oracle.models_ref import PyanNetRef, XVectorSincNetRef  # noqa: E402
from oracle.pyannote_stub import SlidingWindow, SlidingWindowFeature, load_reference_pipelines  # noqa: E402
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream  # noqa: E402

ref = load_reference_pipelines()
seg_net, emb_net = PyanNetRef().eval(), XVectorSincNetRef().eval()
seg_net.load_state_dict(synth_segmentation_state())
emb_net.load_state_dict(synth_embedding_state())
B, S, H = args.batch, 80000, 8000
stream = synth_stream(0, 5.0 + 0.5 * (4 * B - 1))
chunks = [SlidingWindowFeature(stream[i * H:i * H + S, None], SlidingWindow(start=i * 0.5, duration=1 / 16000, step=1 / 16000))
          for i in range((len(stream) - S) // H + 1)]


def pipeline():
    cfg = ref.diarization.SpeakerDiarizationConfig(
        segmentation=ref.models.SegmentationModel(lambda: seg_net), embedding=ref.models.EmbeddingModel(lambda: emb_net),
        device=torch.device("cpu"))
    return ref.diarization.SpeakerDiarization(cfg)


pipe, pos = pipeline(), 0
pipe(chunks[:B])                                       # warm-up batch
pipe, times = pipeline(), []
for r in range(args.repeats):
    if pos + B > len(chunks):
        pipe, pos = pipeline(), 0
    t0 = time.perf_counter()
    out = pipe(chunks[pos:pos + B])
    times.append(time.perf_counter() - t0)
    assert len(out) == B
    pos += B
t = np.array(times)
res = {"what": "reference's own SpeakerDiarization (by path) around the restated networks, Benchmark-shaped batches",
       "host": f"{os.cpu_count()} logical cores, {args.threads} torch threads (build container, not the GPU box)",
       "batch": B, "repeats": args.repeats, "seconds_per_batch_median": round(float(np.median(t)), 4),
       "seconds_per_batch_mean": round(float(t.mean()), 4), "chunks_per_s": round(B / float(np.median(t)), 2),
       "xRT": round(B / float(np.median(t)) / 2, 3)}
# bench.py's CPU leg on the same machine
env = dict(os.environ, OMP_NUM_THREADS=str(args.threads), MKL_NUM_THREADS=str(args.threads), HIP_VISIBLE_DEVICES="")
r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-worker", "--cpu-chunks", str(B), "--cpu-threads", str(args.threads)],
                   capture_output=True, text=True, env=env)
lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
if lines:
    port = json.loads(lines[-1])
    res["bench_cpu_leg_same_machine"] = {k: port[k] for k in ("value", "dedup_value", "repeats", "cores", "kind")}
    res["reference_blocks_over_port"] = round(res["xRT"] / port["value"], 3)
print(json.dumps(res, indent=1))
Path(args.out).write_text(json.dumps(res, indent=1))
