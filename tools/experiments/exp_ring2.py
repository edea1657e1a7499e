#!/usr/bin/env python
"""Ring layout vs dependency: time seg+emb forward passes (no clustering) reading
 (a) a dense resident array view, (b) the ring window with NO pushes in between, (c) the ring with a push before every step
 on one stream, back to back."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import _lib
from diart_amd.models import HipEmbedding, HipSegmentation
from diart_amd.pipeline import AudioRing
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_streams

dev = torch.device("cuda", 0)
n, hop, S = 64, 8000, 80000
audio = torch.from_numpy(synth_streams(n, 30.0, seed0=0)).to(dev)
seg = HipSegmentation(synth_segmentation_state(), max_batch=n).to(dev)
emb = HipEmbedding(synth_embedding_state(), max_batch=n).to(dev)
lib = _lib.load()
hs, he = seg._create(S, n), emb._create(S, n)
out = torch.empty(n, 293, 3, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
ring = AudioRing(n, S, hop, slack_blocks=6, device=dev)
blocks = [audio[:, i * hop:(i + 1) * hop].contiguous() for i in range(40)]
for i in range(10):
    ring.push(blocks[i])


def fwd(base, stride):
    _lib.check(lib.dz_seg_forward(hs, base, stride, n, out.data_ptr(), st))
    _lib.check(lib.dz_emb_frames(he, base, stride, n, st))


def bench(name, fn, reps=30):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per (seg + emb frames)", flush=True)


bench("dense view, same window ", lambda i: fwd(audio.data_ptr(), audio.stride(0)))
bench("dense view, sliding      ", lambda i: fwd(audio.data_ptr() + (i % 20) * hop * 4, audio.stride(0)))
b, s = ring.raw()
bench("ring, no pushes          ", lambda i: fwd(b, s))


def with_push(i):
    ring.push(blocks[10 + i % 25])
    b2, s2 = ring.raw()
    fwd(b2, s2)


bench("ring, push before each   ", with_push)
dense2 = torch.empty(n, 256064, device=dev)
dense2[:, :S] = audio[:, :S]
bench("dense, ring-like pitch   ", lambda i: fwd(dense2.data_ptr(), dense2.stride(0)))
