"""Lifecycle of the engines on the GPU: a service creates and drops `StreamBatch` engines as stream groups come and go
and runs for days (the reference's `StreamingInference` never ends, /root/reference/src/diart/inference.py:101-147).

* engines give their device memory back: the library owns its scratch arenas (hipMalloc behind `dz_seg_create` /
  `dz_emb_create`, not torch's allocator), so a leak would not show in `torch.cuda.memory_allocated`;
* a long run neither grows nor drifts: after thousands of steps the free device memory is where it was after the first
  hundred, and the networks' outputs for a window are bit-identical to what a FRESH engine computes for it."""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MB = 1 << 20


def _free(device):
    torch.cuda.synchronize(device)
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info(device)[0]


def _engine(gpu, n, **kw):
    from diart_amd import models as M
    from diart_amd.pipeline import StreamBatch
    from diart_amd.synth import synth_embedding_state, synth_segmentation_state
    return StreamBatch(M.HipSegmentation(synth_segmentation_state(), max_batch=n),
                       M.HipEmbedding(synth_embedding_state(), max_batch=n), n, device=gpu, tail=True, **kw)


def test_engines_release_their_device_memory(gpu):
    """Create, run and drop an 8-stream engine ten times (two lanes each: handles, arenas, pinned slots, HIP streams).
    The first few engines fill pools that are kept on purpose — torch hands out its 2 x 32 HIP streams round-robin
    (~1 MB of queue memory each) and the HIP runtime grows its per-queue pools for the first ~4 engines (measured:
    -50 / -32 / -32 / -32 MB, then exactly 0 for 26 more engines) — so the statement is about the engines after those:
    from the fifth to the tenth the free device memory does not move."""
    from diart_amd.synth import synth_streams
    audio = torch.from_numpy(synth_streams(8, 7.0, seed0=77)).to(gpu)
    x = torch.zeros(64, device=gpu)
    for i in range(80):                       # torch's stream pools, so that their growth is not counted against the engines
        with torch.cuda.stream(torch.cuda.Stream(gpu, priority=-1 if i % 2 else 0)):
            x.add_(1)
    after = []
    for k in range(10):
        pipe = _engine(gpu, 8)
        for t in range(3):
            pipe(audio[:, t * 8000: t * 8000 + 80000])
        del pipe
        gc.collect()
        after.append(_free(gpu))
    print("free device memory after each engine, MB:", [round(a / MB) for a in after])
    assert max(after[5:]) - min(after[5:]) < 8 * MB, [round(a / MB) for a in after]
    assert after[0] - after[-1] < 512 * MB, [round(a / MB) for a in after]          # the pools themselves stay small


def test_soak_memory_flat_and_no_drift(gpu):
    """3 000 steps of a 16-stream engine over a looping 60 s corpus (its clustering states simply keep running): free
    device memory flat after the first hundred steps, no range flag, no NaN, and the last step's segmentation /
    embeddings bit-identical to a fresh engine's on the same windows."""
    from diart_amd.synth import synth_streams
    n, hop, S = 16, 8000, 80000
    audio = torch.from_numpy(synth_streams(n, 60.0, seed0=500)).to(gpu)
    T = (audio.shape[1] - S) // hop
    pipe = _engine(gpu, n)
    inflight, free100, seg_last, emb_last = [], None, None, None
    steps = 3000
    for t in range(steps):
        w = (t % T) * hop
        inflight.append(pipe.launch(audio[:, w: w + S]))
        if len(inflight) >= pipe.max_inflight:
            seg, emb, scores, assign = pipe.finish(inflight.pop(0))
            assert np.isfinite(seg).all()
        if t == 100:
            free100 = torch.cuda.mem_get_info(gpu)[0]
    while inflight:
        seg, emb, scores, assign = pipe.finish(inflight.pop(0))
        seg_last, emb_last = seg.copy(), emb.copy()
    free_end = torch.cuda.mem_get_info(gpu)[0]
    assert abs(free_end - free100) < 16 * MB, (free100 / MB, free_end / MB)
    fresh = _engine(gpu, n)
    w = ((steps - 1) % T) * hop
    seg0, emb0, _, _ = fresh(audio[:, w: w + S])
    assert np.array_equal(seg0, seg_last)
    both_nan = np.isnan(emb0) & np.isnan(emb_last)
    assert np.array_equal(np.where(both_nan, 0, emb0), np.where(both_nan, 0, emb_last))


def _run_alone(gpu, audio, steps):
    pipe = _engine(gpu, audio.shape[0])
    out = []
    for t in range(steps):
        seg, emb, scores, assign = pipe(audio[:, t * 8000: t * 8000 + 80000])
        out.append((seg.copy(), emb.copy(), scores.copy(), assign.copy()))
    return out


def _same(a, b):
    for x, y in zip(a, b):
        both = np.isnan(x) & np.isnan(y) if x.dtype.kind == "f" else np.zeros(x.shape, bool)
        if not np.array_equal(np.where(both, 0, x), np.where(both, 0, y)):
            return False
    return True


def test_two_engines_interleaved_and_on_two_threads(gpu):
    """Two engines in one process (two stream groups of a service, /root/reference/src/diart/sources.py:204-271 has one
    source per connection): (1) their steps interleaved on one thread with both kept in flight, (2) each driven by its
    own thread at the same time (ctypes releases the GIL: the library's error slot, range flag, context scratch and
    host pool see real concurrency).  Every step's segmentation, embeddings, scores and assignments equal what each
    engine computes alone."""
    import threading
    from diart_amd.synth import synth_streams
    steps = 12
    audio = [torch.from_numpy(synth_streams(8, 5.0 + 0.5 * steps, seed0=900 + 50 * k)).to(gpu) for k in range(2)]
    alone = [_run_alone(gpu, a, steps) for a in audio]

    pipes = [_engine(gpu, 8) for _ in range(2)]
    tickets, got = [[], []], [[], []]
    for t in range(steps):
        for k in range(2):
            tickets[k].append(pipes[k].launch(audio[k][:, t * 8000: t * 8000 + 80000]))
        for k in range(2):
            if len(tickets[k]) >= 2:
                got[k].append(tuple(np.copy(v) for v in pipes[k].finish(tickets[k].pop(0))))
    for k in range(2):
        while tickets[k]:
            got[k].append(tuple(np.copy(v) for v in pipes[k].finish(tickets[k].pop(0))))
        assert len(got[k]) == steps and all(_same(g, a) for g, a in zip(got[k], alone[k])), f"interleaved, engine {k}"
    del pipes

    res, errs = [None, None], []

    def drive(k):
        try:
            res[k] = _run_alone(gpu, audio[k], steps)
        except Exception as exc:      # noqa: BLE001
            errs.append((k, repr(exc)))

    for rep in range(3):
        th = [threading.Thread(target=drive, args=(k,)) for k in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
        for k in range(2):
            assert all(_same(g, a) for g, a in zip(res[k], alone[k])), f"threads, repetition {rep}, engine {k}"


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_very_large_batches_are_right_or_loud(gpu, precision):
    """`max_batch` has no upper bound in the C ABI and a reference-shaped caller may ask for anything
    (`Benchmark(batch_size=...)`, /root/reference/src/diart/inference.py:275; the embedding model sees batch x speakers
    rows, blocks/embedding.py:57-59).  Kernels address their operands with 32-bit offsets, so a large batch must either
    compute what the same chunks give in batches of 64 (segmentation bit for bit, embeddings to 1e-6: the library's
    own batch-independence property) or fail with an error — never return garbage.  1 024 and 4 000 chunks (the LSTM's
    x-projection of 4 000 chunks is 4.8 GB, past every 32-bit byte offset; measured beyond that: 8 000 chunks are refused
    by the split-f16 embedding and computed by the exact-f32 path through its round-1 kernels, 2.8e-6 from the batches
    of 64)."""
    from diart_amd import models as M
    from diart_amd._lib import DiartAmdError
    from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream
    stream = torch.from_numpy(synth_stream(33, 5.0 + 0.05 * 4000 + 1.0)).to(gpu)
    seg = M.HipSegmentation(synth_segmentation_state(), max_batch=64, precision=precision).to(gpu)
    emb = M.HipEmbedding(synth_embedding_state(), max_batch=64, precision=precision).to(gpu)
    for B in (1024, 4000):
        x = stream.unfold(0, 80000, 800)[:B][:, None, :]          # B overlapping windows, read in place
        assert x.shape[0] == B
        w = (torch.rand(B, 3, 293, generator=torch.Generator().manual_seed(B)) ** 2 + 1e-8).to(gpu)
        ref_s = torch.cat([seg(x[i:i + 64]) for i in range(0, B, 64)]).cpu()
        ref_e = torch.cat([emb.forward_multi(x[i:i + 64], w[i:i + 64]) for i in range(0, B, 64)]).cpu()
        try:
            got_s = seg(x).cpu()
        except DiartAmdError as exc:
            print(f"segmentation, {B} chunks: refused ({str(exc)[:120]})")
        else:
            assert torch.equal(got_s, ref_s), f"segmentation at batch {B} differs from batches of 64"
        try:
            got_e = emb.forward_multi(x, w).cpu()
        except DiartAmdError as exc:
            print(f"embedding, {B} chunks: refused ({str(exc)[:120]})")
        else:
            rel = ((got_e - ref_e).norm(dim=-1) / ref_e.norm(dim=-1)).max().item()
            assert rel < 1e-6, f"embedding at batch {B}: relative difference {rel} to batches of 64"
        # the models go back to small batches afterwards
        assert torch.equal(seg(x[:3]).cpu(), ref_s[:3])
