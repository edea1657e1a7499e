#!/usr/bin/env python
"""Where does the host-fed pass lose time?  Variants of the bench loop on one GPU:
A resident windows, launches on the default stream (= bench `value`)
B resident windows, launches inside a side stream
C device ring fed from DEVICE blocks (no PCIe)
D device ring fed from pinned HOST blocks (= bench `host_fed`)
E as D with the push of step t+1 issued BEFORE finish(t - depth) (one step earlier)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd.models import HipEmbedding, HipSegmentation
from diart_amd.pipeline import AudioRing, StreamBatch
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_streams

dev = torch.device("cuda", 0)
n, hop, S, steps, warm = 64, 8000, 80000, 150, 10
total = steps + warm
audio_cpu = torch.from_numpy(synth_streams(n, (S + hop * (total + 2)) / 16000.0, seed0=0))
audio = audio_cpu.to(dev)
pipe = StreamBatch(HipSegmentation(synth_segmentation_state(), max_batch=n), HipEmbedding(synth_embedding_state(), max_batch=n),
                   n, device=dev, tail=True)
blocks = audio_cpu.unfold(1, hop, hop)
pinned = [blocks[:, i].contiguous().pin_memory() for i in range(S // hop + total)]
dblocks = [p.to(dev) for p in pinned]
feed = torch.cuda.Stream(dev)


def loop(kind, first, count, ring=None):
    inflight = []
    for t in range(first, first + count):
        if kind == "A":
            inflight.append(pipe.launch(audio[:, t * hop: t * hop + S]))
        elif kind == "B":
            with torch.cuda.stream(feed):
                inflight.append(pipe.launch(audio[:, t * hop: t * hop + S]))
        else:
            src = dblocks if kind == "C" else pinned
            with torch.cuda.stream(feed):
                ring.push(src[S // hop - 1 + t])
                inflight.append(pipe.launch(ring))
        if len(inflight) > pipe.depth:
            pipe.finish(inflight.pop(0), want_scores=True)
    while inflight:
        pipe.finish(inflight.pop(0), want_scores=True)


for kind in ["A", "B", "C", "D", "A", "D", "C", "B"]:
    ring = None
    if kind in "CD":
        ring = AudioRing(n, S, hop, slack_blocks=2 * pipe.depth + 2, device=dev)
        src = dblocks if kind == "C" else pinned
        with torch.cuda.stream(feed):
            for i in range(S // hop - 1):
                ring.push(src[i])
    loop(kind, 0, warm, ring)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(kind, warm, steps, ring)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{kind}: {1e3 * dt / steps:.3f} ms/step  {n * steps / dt / 2:.0f} xRT", flush=True)
