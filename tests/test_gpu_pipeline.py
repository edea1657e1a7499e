"""End-to-end parity of the hot path on the GPU (segmentation -> OSP -> embedding -> normalisation
-> clustering through StreamBatch and through the blocks) against the CPU oracle, plus the
size-independent properties that stand in for the oracle at BASELINE.json's full size
(64 concurrent streams): batch invariance, in-place rolling window == copy, permutation
equivariance, run-to-run determinism, reference-shaped (B*K rows) call == de-duplicated call.
"""
import os

import numpy as np
import pytest
import torch

from diart_amd import _lib
from diart_amd import models as M
from diart_amd.blocks import (OnlineSpeakerClustering, OverlapAwareSpeakerEmbedding,
                              SpeakerSegmentation)
from diart_amd.features import SlidingWindow, SlidingWindowFeature
from diart_amd.pipeline import StreamBatch
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_streams

pytestmark = pytest.mark.gpu

SEG_MAX, EMB_COS = 1e-4, 0.99999


@pytest.fixture(scope="module")
def oracle_models():
    from oracle.models_ref import PyanNetRef, XVectorSincNetRef
    s, e = PyanNetRef().eval(), XVectorSincNetRef().eval()
    s.load_state_dict(synth_segmentation_state())
    e.load_state_dict(synth_embedding_state())
    return s, e


def test_stream_batch_matches_oracle(gpu, oracle_models):
    """4 streams x 14 steps.  Networks within the stated tolerances; clustering BIT-EXACT when the
    oracle clustering is fed the GPU's segmentation / embeddings (same inputs -> same fp64
    arithmetic), and the fully-CPU chain (oracle nets -> oracle clustering) must agree on the
    speaker assignments except where a decision sits inside fp32 noise."""
    from oracle.clustering_ref import OnlineSpeakerClusteringRef
    from oracle.functional_ref import normalize_embeddings_ref, overlapped_speech_penalty_ref
    n, steps = 4, 14
    audio = torch.from_numpy(synth_streams(n, 5.0 + 0.5 * steps, seed0=300))
    d_audio = audio.to(gpu)
    pipe = StreamBatch(M.HipSegmentation(synth_segmentation_state(), max_batch=n),
                       M.HipEmbedding(synth_embedding_state(), max_batch=n), n, device=gpu)
    clu_same = [OnlineSpeakerClusteringRef(0.6, 0.3, 1.0, "cosine", 20) for _ in range(n)]
    clu_cpu = [OnlineSpeakerClusteringRef(0.6, 0.3, 1.0, "cosine", 20) for _ in range(n)]
    agree = total = 0
    for t in range(steps):
        win = slice(t * 8000, t * 8000 + 80000)
        seg, emb, scores, assign = pipe(d_audio[:, win])
        with torch.no_grad():
            x = audio[:, None, win]
            rseg = oracle_models[0](x)
            remb = normalize_embeddings_ref(
                oracle_models[1].forward_multi(x, overlapped_speech_penalty_ref(rseg)))
        assert np.abs(seg - rseg.numpy()).max() < SEG_MAX
        cos = (torch.from_numpy(emb) * remb).sum(-1)
        assert cos.min().item() > EMB_COS
        for i in range(n):
            want, want_assign = clu_same[i](seg[i], emb[i])
            assert np.array_equal(scores[i], want)
            assert np.array_equal(assign[i], np.asarray(want_assign))
            _, cpu_assign = clu_cpu[i](rseg[i].numpy(), remb[i].numpy())
            agree += int(np.sum(assign[i] == np.asarray(cpu_assign)))
            total += assign[i].size
    assert agree / total >= 0.98, f"assignment agreement with the all-CPU chain {agree}/{total}"


def test_blocks_keep_reference_types_and_shapes(gpu):
    """SpeakerSegmentation / OverlapAwareSpeakerEmbedding / OnlineSpeakerClustering through the
    reference's block API: input kind is restored, outputs live on the host, B=1 keeps the
    (1,K,D) shape after normalisation (blocks/embedding.py:68 + functional.py:20-21)."""
    seg_model = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=4)
    emb_model = M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=4)
    seg_block = SpeakerSegmentation(seg_model, gpu)
    emb_block = OverlapAwareSpeakerEmbedding(emb_model, gamma=3, beta=10, norm=1, device=gpu)
    audio = synth_streams(1, 6.0, seed0=9)[0]
    sw = SlidingWindow(start=0.0, duration=1 / 16000, step=1 / 16000)
    chunk = SlidingWindowFeature(audio[:80000, None], sw)
    seg = seg_block(chunk)
    assert isinstance(seg, SlidingWindowFeature) and seg.data.shape == (293, 3)
    seg_np = seg_block(audio[None, :80000, None])
    assert isinstance(seg_np, np.ndarray) and seg_np.shape == (1, 293, 3)
    assert np.array_equal(seg_np[0], seg.data)
    batch = torch.from_numpy(np.stack([audio[:80000], audio[8000:88000]]))[:, :, None]
    seg_t = seg_block(batch)
    assert isinstance(seg_t, torch.Tensor) and seg_t.device.type == "cpu" and seg_t.shape == (2, 293, 3)
    emb = emb_block(batch, seg_t)
    assert emb.shape == (2, 3, 512) and emb.device.type == "cpu"
    assert torch.allclose(emb.norm(dim=-1), torch.ones(2, 3), atol=1e-5)
    emb1 = emb_block(batch[:1], seg_t[:1])
    assert emb1.shape == (1, 3, 512)
    assert (emb1[0] - emb[0]).abs().max().item() < 1e-6
    clu = OnlineSpeakerClustering(0.6, 0.3, 1.0, "cosine", 20)
    frames = SlidingWindow(start=0.0, duration=5 / 293, step=5 / 293)
    out = clu(SlidingWindowFeature(seg_t[0].numpy(), frames), emb[0])
    assert isinstance(out, SlidingWindowFeature) and out.data.shape == (293, 20)
    assert out.data.dtype == np.float64
    with pytest.raises(ValueError):
        seg_block("not a feature")


def test_full_size_properties_64_streams(gpu):
    """BASELINE.json config 2 size (64 chunks per launch); the oracle would need minutes here, so
    the gate is structural: every property below holds bit-exactly because each chunk's
    arithmetic is independent of its position in the batch."""
    n = 64
    seg = M.HipSegmentation(synth_segmentation_state(), max_batch=n).to(gpu)
    emb = M.HipEmbedding(synth_embedding_state(), max_batch=3 * n).to(gpu)
    audio = torch.from_numpy(synth_streams(n, 5.5, seed0=500)).to(gpu)     # (64, 88000)
    view = audio[:, 8000:88000]                                            # in-place window
    assert not view.is_contiguous()
    s_all = seg(view[:, None, :])
    assert s_all.shape == (n, 293, 3) and torch.isfinite(s_all).all()
    assert 0.0 <= s_all.min().item() and s_all.max().item() <= 1.0
    # in-place strided window == contiguous copy; run-to-run determinism
    assert torch.equal(s_all, seg(view.contiguous()[:, None, :]))
    assert torch.equal(s_all, seg(view[:, None, :]))
    # batch invariance: 8 slices of 8 == one launch of 64
    for i in range(0, n, 8):
        assert torch.equal(s_all[i:i + 8], seg(view[i:i + 8, None, :]))
    # permutation equivariance
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(gpu)
    assert torch.equal(seg(view.contiguous()[perm][:, None, :]), s_all[perm])
    # embeddings: de-duplicated call == reference-shaped (B*K repeated rows) call, per slice
    from diart_amd.functional import overlapped_speech_penalty
    w = overlapped_speech_penalty(s_all, 3, 10, speaker_major=True)        # (64,3,293)
    e_all = emb.forward_multi(view[:, None, :], w, normalize=True)
    assert e_all.shape == (n, 3, 512) and torch.isfinite(e_all).all()
    assert torch.allclose(e_all.norm(dim=-1), torch.ones(n, 3, device=gpu), atol=1e-5)
    rows = view.contiguous()[:, None, :].repeat(1, 3, 1).reshape(3 * n, 1, -1)
    e_rows = emb(rows, w.reshape(3 * n, 293))
    e_rows = e_rows / e_rows.norm(dim=-1, keepdim=True)
    assert (e_rows.view(n, 3, 512) - e_all).abs().max().item() < 2e-6
    # batch invariance of the embeddings.  The frame features are bit-identical whatever the batch
    # (as for the segmentation above); the statistics pooling fused into tdnn5's epilogue (round 3)
    # merges per-TILE moments, and which 128-row tiles a chunk's frames fall into depends on the
    # chunk's position in the flattened batch: the same embedding to a few f32 ulps (deterministic for a
    # given batch), bit-identical again on the two-launch path (option pool_fuse = 0)
    for i in range(0, n, 16):
        part = emb.forward_multi(view[i:i + 16, None, :], w[i:i + 16], normalize=True)
        assert (e_all[i:i + 16] - part).abs().max().item() < 5e-7
    assert torch.equal(e_all, emb.forward_multi(view[:, None, :], w, normalize=True))     # run-to-run determinism
    _lib.set_option("pool_fuse", 0)
    try:
        e_two = emb.forward_multi(view[:, None, :], w, normalize=True)
        for i in range(0, n, 16):
            assert torch.equal(e_two[i:i + 16], emb.forward_multi(view[i:i + 16, None, :], w[i:i + 16], normalize=True))
        assert (e_two - e_all).abs().max().item() < 5e-7
    finally:
        _lib.set_option("pool_fuse", 1)
    # different streams give different embeddings (the batch is not aliased)
    assert (e_all[0] - e_all[1]).abs().max().item() > 1e-3


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("recurrence", ["valu", "4"])
def test_full_size_invariances_of_the_graphs(gpu, precision, recurrence):
    """More oracle-free properties at BASELINE.json config 2 size (64 chunks per launch), this time of the GRAPHS
    (third-party PyanNet / XVectorSincNet reached from /root/reference/src/diart/models.py:133, :262), in both
    arithmetic modes and with the latency- and the throughput-form recurrence:

    * both networks start with InstanceNorm1d(1) on the waveform: a DC offset changes nothing but roundings;
    * the statistics pooling is a ratio of weighted sums (paper Eq. 1): scaling the pooling weights of a speaker by
      a power of two scales every sum exactly — the embedding does not move by a bit;
    * the K speakers of a chunk are pooled independently from shared frame features: permuting the weight rows
      permutes the embeddings, bit for bit;
    * an all-zero weight row is a speaker that is never active: the pooling has nothing to average, the row is NaN (torch's
      0 / 0) and the other speakers of the chunk are untouched."""
    if recurrence != "valu" and precision == "f32":
        pytest.skip("the matrix-core recurrences are split-f16 kernels")
    n = 64
    seg = M.HipSegmentation(synth_segmentation_state(), max_batch=n, precision=precision, recurrence=recurrence).to(gpu)
    emb = M.HipEmbedding(synth_embedding_state(), max_batch=n, precision=precision).to(gpu)
    x = torch.from_numpy(synth_streams(n, 5.0, seed0=900)).to(gpu)[:, None, :80000]
    s0 = seg(x)
    s_dc = seg(x + 0.25)
    assert (s_dc - s0).abs().max().item() < 5e-5, (s_dc - s0).abs().max().item()
    from diart_amd.functional import overlapped_speech_penalty
    w = overlapped_speech_penalty(s0, 3, 10, speaker_major=True).contiguous()          # (64, 3, 293)
    e0 = emb.forward_multi(x, w, normalize=True)
    e_dc = emb.forward_multi(x + 0.25, w, normalize=True)
    assert (e_dc * e0).sum(-1).min().item() > EMB_COS
    # weight scale (per speaker, powers of two)
    scale = torch.tensor([4.0, 0.5, 1024.0], device=gpu)[None, :, None]
    e_sc = emb.forward_multi(x, (w * scale).contiguous(), normalize=True)
    assert torch.equal(e_sc, e0)
    # speaker permutation
    perm = [2, 0, 1]
    e_pm = emb.forward_multi(x, w[:, perm].contiguous(), normalize=True)
    assert torch.equal(e_pm, e0[:, perm])
    # a speaker without weight
    wz = w.clone()
    wz[5, 1] = 0.0
    e_z = emb.forward_multi(x, wz, normalize=True)
    assert torch.isnan(e_z[5, 1]).all()
    keep = torch.ones(n, 3, dtype=torch.bool, device=gpu)
    keep[5, 1] = False
    assert torch.equal(e_z[keep], e0[keep])


def test_audio_ring_is_the_rolling_window(gpu):
    """dz_ring_*: pushing 500 ms blocks reproduces the windows rearrange_audio_stream emits
    (operators.py:44-100: the last `duration` seconds after every block, from the first complete
    window on), from pinned host blocks and from device blocks, across several wrap-arounds."""
    from diart_amd.pipeline import AudioRing
    n, W, hop, steps = 3, 80000, 8000, 37
    audio = torch.from_numpy(synth_streams(n, (W + hop * steps) / 16000.0, seed0=40))
    ring = AudioRing(n, W, hop, slack_blocks=2, device=gpu)
    emitted = 0
    for b in range(W // hop + steps):
        blk = audio[:, b * hop:(b + 1) * hop]
        blk = blk.contiguous().pin_memory() if b % 2 == 0 else blk.to(gpu)
        full = ring.push(blk)
        assert full == (b + 1 >= W // hop)
        if full:
            t = b + 1 - W // hop
            assert torch.equal(ring.snapshot().cpu(), audio[:, t * hop:t * hop + W]), f"window {t}"
            emitted += 1
    assert emitted == steps + 1
    ring.reset()
    assert ring.filled == 0


@pytest.mark.parametrize("na,nb", [(64 * 293 * 3, 64 * 3 * 512), (879, 1536), (5, 0), (4, 3), (1, 1), (1027, 2)])
def test_results_to_host_is_an_exact_copy(gpu, na, nb):
    """dz_results_to_host (the `.cpu()` of blocks/segmentation.py:47 and blocks/embedding.py:68 for a whole step):
    two device buffers -> two pinned host buffers in one launch, bit-exact, any length (16-byte pieces + a scalar
    tail), nothing written behind the end; misaligned buffers are refused.  StreamBatch uses it instead of
    hipMemcpyAsync (csrc/ring.hip: that call blocks now and then)."""
    from diart_amd import _lib
    lib, ctx = _lib.load(), _lib.context(gpu.index or 0)
    g = torch.Generator().manual_seed(na + 7 * nb)
    a = torch.randn(na + 8, generator=g).to(gpu)
    b = torch.randn(max(nb, 1) + 8, generator=g).to(gpu)
    ha = torch.full((na + 8,), -7.0).pin_memory()
    hb = torch.full((max(nb, 1) + 8,), -7.0).pin_memory()
    st = torch.cuda.current_stream(gpu)
    _lib.check(lib.dz_results_to_host(ctx, a.data_ptr(), ha.data_ptr(), na, b.data_ptr() if nb else None,
                                      hb.data_ptr() if nb else None, nb, st.cuda_stream))
    ev = torch.cuda.Event()
    ev.record(st)
    ev.synchronize()
    assert torch.equal(ha[:na], a[:na].cpu()) and bool((ha[na:] == -7.0).all())
    assert torch.equal(hb[:nb], b[:nb].cpu()) and bool((hb[nb:] == -7.0).all())
    assert lib.dz_results_to_host(ctx, a.data_ptr() + 4, ha.data_ptr(), 4, None, None, 0, st.cuda_stream) != 0


def test_stream_batch_results_by_kernel_equal_the_memcpy_pair(gpu):
    """The two ways of bringing a step's results to the host give the same arrays."""
    n, S, hop, steps = 8, 80000, 8000, 4
    audio = torch.from_numpy(synth_streams(n, (S + hop * steps) / 16000.0, seed0=91)).to(gpu)
    outs = []
    for by_kernel in (True, False):
        pipe = StreamBatch(M.HipSegmentation(synth_segmentation_state(), max_batch=n), M.HipEmbedding(synth_embedding_state(), max_batch=n),
                           n, device=gpu, tail=False)
        pipe.d2h_by_kernel = by_kernel
        got = []
        for t in range(steps):
            seg, emb, _, assign = pipe.finish(pipe.launch(audio[:, t * hop: t * hop + S]))
            got.append((seg.copy(), emb.copy(), assign.copy()))
        outs.append(got)
    for (s0, e0, a0), (s1, e1, a1) in zip(*outs):
        assert np.array_equal(s0, s1) and np.array_equal(e0, e1, equal_nan=True) and np.array_equal(a0, a1)


def test_stream_batch_on_ring_with_splits_and_tail(gpu):
    """StreamBatch reading the device ring, with the networks split into sub-batches on separate
    HIP streams, and the C++ output tail: identical segmentation / embeddings / clustering to the
    plain resident-audio single-batch run, and the speech turns equal the Python
    DelayedAggregation + Binarize blocks (pinned to the reference's goldens) on the same scores."""
    from diart_amd.blocks import Binarize, DelayedAggregation
    from diart_amd.pipeline import AudioRing
    n, W, hop, steps, latency = 6, 80000, 8000, 9, 1.5
    audio = torch.from_numpy(synth_streams(n, (W + hop * steps) / 16000.0, seed0=700))
    d_audio = audio.to(gpu)
    seg_sd, emb_sd = synth_segmentation_state(), synth_embedding_state()
    plain = StreamBatch(M.HipSegmentation(seg_sd, max_batch=n), M.HipEmbedding(emb_sd, max_batch=n), n,
                        device=gpu, seg_split=1, emb_split=1)
    split = StreamBatch(M.HipSegmentation(seg_sd, max_batch=n), M.HipEmbedding(emb_sd, max_batch=n), n,
                        device=gpu, seg_split=3, emb_split=2, tail=True, latency=latency)
    ring = AudioRing(n, W, hop, device=gpu)
    for b in range(W // hop - 1):
        ring.push(audio[:, b * hop:(b + 1) * hop].contiguous().pin_memory())
    aggs = [DelayedAggregation(0.5, latency, "hamming", "loose") for _ in range(n)]
    binarize = Binarize(0.6)
    bufs = [[] for _ in range(n)]
    res = 5.0 / 293
    for t in range(steps):
        b = W // hop - 1 + t
        assert ring.push(audio[:, b * hop:(b + 1) * hop].contiguous().pin_memory())
        seg0, emb0, sc0, as0 = plain(d_audio[:, t * hop:t * hop + W])
        ticket = split.launch(ring)
        seg1, emb1, sc1, as1 = split.finish(ticket)
        # (the 6-chunk launch pools inside tdnn5's epilogue, the 3-chunk sub-batches are in the latency
        # regime and pool in a launch of their own: the same embeddings to a few f32 ulps)
        assert np.array_equal(seg0, seg1) and np.abs(emb0 - emb1).max() < 5e-7
        assert np.array_equal(sc0, sc1) and np.array_equal(as0, as1)
        agg, rows, t0, r, turns, nturns = ticket["tail"]
        for i in range(n):
            bufs[i].append(SlidingWindowFeature(sc1[i], SlidingWindow(start=t * 0.5, duration=res, step=res)))
            want = aggs[i](bufs[i])
            assert np.array_equal(agg[i, :rows[i]], want.data)
            assert abs(t0[i] - want.sliding_window.start) < 1e-12 and abs(r[i] - want.sliding_window.step) < 1e-12
            wt = sorted((s.start, s.end, float(k)) for s, k, _ in binarize(want).itertracks(yield_label=True))
            got = sorted(map(tuple, turns[i, :nturns[i]]))
            assert np.allclose(np.array(got).reshape(-1, 3), np.array(wt).reshape(-1, 3), rtol=0, atol=1e-12)
            if len(bufs[i]) == aggs[i].num_overlapping_windows:
                bufs[i] = bufs[i][1:]


def test_ring_rows_advance_independently(gpu):
    """dz_ring_push_rows / dz_ring_gather: every stream of the ring has its own write position; the
    gathered windows equal a host-side rolling window of each stream, whatever the interleaving."""
    from diart_amd.pipeline import AudioRing
    n, W, hop = 5, 800, 80
    ring = AudioRing(n, W, hop, slack_blocks=0, device=gpu)
    rng = np.random.default_rng(11)
    hist = [np.zeros(0, dtype=np.float32) for _ in range(n)]
    out = torch.empty(n, W, device=gpu)
    stage = torch.empty(n, hop).pin_memory()
    for it in range(60):
        rows = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
        blk = rng.standard_normal((len(rows), hop)).astype(np.float32)
        torch.cuda.synchronize()                # the previous round's zero-copy read of `stage` is over
        stage[:len(rows)].copy_(torch.from_numpy(blk))
        ring.push_rows(stage[:len(rows)], rows)
        for j, r in enumerate(rows):
            hist[r] = np.concatenate([hist[r], blk[j]])[-W:]
        full = [r for r in range(n) if len(hist[r]) == W]
        assert all(ring.filled_row(r) == min(W, len(hist[r])) for r in range(n))
        if full:
            got = ring.gather(full, out).cpu().numpy()
            assert np.array_equal(got, np.stack([hist[r] for r in full]))
    ring.reset_row(2)
    assert ring.filled_row(2) == 0 and ring.filled_row(1) == W
    with pytest.raises(RuntimeError):
        ring.gather([2], out)                   # incomplete window: refused, not garbage


@pytest.mark.parametrize("device_rings", [True, False])
def test_stream_server_equals_dedicated_pipelines(gpu, device_rings):
    """diart_amd.serve.StreamServer: 4 streams of different lengths that join at different times and
    push audio in odd block sizes, windows batched ACROSS streams; every stream's accumulated RTTM is
    exactly what its own SpeakerDiarization pipeline (blocks API, batch 1, same models) produces,
    for latency = step and for a longer latency (aggregation over 3 windows)."""
    from diart_amd.blocks import SpeakerDiarization, SpeakerDiarizationConfig
    from diart_amd.inference import StreamingInference
    from diart_amd.serve import StreamServer
    seg_sd, emb_sd = synth_segmentation_state(), synth_embedding_state()
    lengths = {"alice": 11.0, "bob": 8.5, "carol": 14.0, "dave": 6.0}
    audio = {k: synth_streams(1, v, seed0=900 + i)[0] for i, (k, v) in enumerate(lengths.items())}
    for latency in (0.5, 1.5):
        srv = StreamServer(M.HipSegmentation(seg_sd, max_batch=4), M.HipEmbedding(emb_sd, max_batch=4),
                           max_streams=4, latency=latency, device=gpu, device_rings=device_rings)
        assert (srv.rings is not None) == device_rings
        rng = np.random.default_rng(3)
        pos = {k: 0 for k in audio}
        join_at = {"alice": 0, "bob": 3, "carol": 3, "dave": 9}     # server ticks at which they open
        tick, widths = 0, []
        while any(pos[k] < len(audio[k]) for k in audio):
            for k in audio:
                if tick == join_at[k]:
                    srv.open(k)
                if tick >= join_at[k] and pos[k] < len(audio[k]):
                    n = int(rng.integers(2000, 30000))
                    srv.push(k, audio[k][pos[k]:pos[k] + n])
                    pos[k] += n
            out = srv.step()
            widths.append(len(out))
            tick += 1
        srv.drain()
        assert max(widths) >= 3, "windows of different streams were never batched together"
        for k in audio:
            got = srv.close(k)
            cfg = SpeakerDiarizationConfig(
                segmentation=M.SegmentationModel.from_state(seg_sd, max_batch=1),
                embedding=M.EmbeddingModel.from_state(emb_sd, max_batch=1), latency=latency, device=gpu)
            usable = len(audio[k]) // 8000 * 8000       # the server never sees a zero-padded last block
            want = StreamingInference(SpeakerDiarization(cfg), audio[k][:usable], 16000, k, (0, 0), 1)()
            assert want is not None and got.to_rttm() == want.to_rttm(), (k, latency)


def test_websocket_front_end_streams_rttm_from_the_gpu(gpu):
    """diart_amd.ws.WebSocketFrontEnd over real sockets on top of a GPU StreamServer: two clients send
    the reference's message format (base64 float32 text); the RTTM lines each gets back are, in order,
    exactly the per-chunk ``Annotation.to_rttm()`` lines (console/serve.py:124) of an identical
    StreamServer stepped directly with the same audio."""
    import sys
    import time
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_ws import Client
    from diart_amd.serve import StreamServer
    from diart_amd.ws import WebSocketFrontEnd
    seg_sd, emb_sd = synth_segmentation_state(), synth_embedding_state()
    audio = {k: synth_streams(1, 9.0, seed0=700 + i)[0] for i, k in enumerate(("ann", "ben"))}

    def make():
        return StreamServer(M.HipSegmentation(seg_sd, max_batch=2), M.HipEmbedding(emb_sd, max_batch=2),
                            max_streams=2, device=gpu)

    direct, want = make(), {k: [] for k in audio}
    for k in audio:
        direct.open(k)
        direct.push(k, audio[k])
    while True:
        out = direct.step()
        if not out:
            break
        for k, ann in out.items():
            want[k] += [l for l in ann.to_rttm().splitlines() if l]
    assert all(want.values())

    srv = make()
    fe = WebSocketFrontEnd(srv, port=0).start()
    try:
        clients = {k: Client(fe.port, k) for k in audio}
        for k, c in clients.items():
            for pos in range(0, len(audio[k]), 16000):
                c.send_audio(audio[k][pos:pos + 16000])
        deadline = time.time() + 30
        while time.time() < deadline and sum(s.emitted for s in list(srv._streams.values())) < 18:
            time.sleep(0.05)
        assert sum(s.emitted for s in srv._streams.values()) == 18      # 9 s = 9 windows per stream
        time.sleep(0.3)
        got = {k: [] for k in audio}
        for k, c in clients.items():
            c.s.settimeout(0.5)
            try:
                while True:
                    op, data = c.recv()
                    assert op == 0x1
                    got[k] += [l for l in data.decode().splitlines() if l]
            except (TimeoutError, OSError):
                pass
        assert not fe.errors
        assert got == want
    finally:
        fe.stop()


def test_front_half_on_its_own_stream_is_the_same_arithmetic(gpu, monkeypatch):
    """DZ_SEG_FRONT=1: SincNet + the first x-projection of step t + depth on their own stream under the
    recurrences of step t (dz_seg_front / dz_seg_back).  Same kernels, same operands: bit-identical
    outputs, with depth + 1 steps in flight (the front half of a lane's next step overlaps its current
    back half; the handle's event guards the one buffer they share)."""
    if not _lib.experiments():
        pytest.skip("DZ_SEG_FRONT is honoured by the experiments build only")
    n, W, hop, steps = 8, 80000, 8000, 9
    audio = torch.from_numpy(synth_streams(n, (W + hop * steps) / 16000.0, seed0=910)).to(gpu)
    seg_sd, emb_sd = synth_segmentation_state(), synth_embedding_state()

    def run(front, split=1):
        monkeypatch.setenv("DZ_SEG_FRONT", "1" if front else "0")
        sb = StreamBatch(M.HipSegmentation(seg_sd, max_batch=n), M.HipEmbedding(emb_sd, max_batch=n), n,
                         device=gpu, seg_split=split, emb_split=1)
        assert sb.seg_front == front
        inflight, out = [], []
        for t in range(steps):
            inflight.append(sb.launch(audio[:, t * hop:t * hop + W]))
            if len(inflight) > sb.depth:
                out.append([np.array(x) for x in sb.finish(inflight.pop(0))])
        while inflight:
            out.append([np.array(x) for x in sb.finish(inflight.pop(0))])
        return out

    want = run(False)
    for split in (1, 2):
        got = run(True, split)
        for t in range(steps):
            for a, b in zip(want[t], got[t]):
                assert np.array_equal(a, b), (split, t)


def test_back_half_refuses_a_batch_the_front_half_did_not_prepare(gpu):
    from diart_amd import _lib
    lib = _lib.load()
    seg = M.HipSegmentation(synth_segmentation_state(), max_batch=4).to(gpu)
    h = seg._create(80000, 4)
    try:
        x = torch.from_numpy(synth_streams(4, 5.0, seed0=5)).to(gpu)
        out = torch.empty((4, 293, 3), device=gpu)
        st = torch.cuda.current_stream(gpu).cuda_stream
        assert lib.dz_seg_back(h, 4, out.data_ptr(), 3.0, 10.0, 0, None, st) != 0     # nothing prepared
        assert b"dz_seg_front prepared 0" in lib.dz_last_error()
        assert lib.dz_seg_front(h, x.data_ptr(), x.stride(0), 3, st) == 0
        assert lib.dz_seg_back(h, 4, out.data_ptr(), 3.0, 10.0, 0, None, st) != 0
        assert lib.dz_seg_back(h, 3, out.data_ptr(), 3.0, 10.0, 0, None, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(out[:3], seg(x[:3, None, :]))
    finally:
        seg._destroy(h)


def test_warm_steps_leave_no_trace(gpu, monkeypatch):
    """StreamBatch runs warm steps on silence before the first real one (pipeline.py _warm_up: the output
    tail is built and the first-launch set-up absorbed while no stream is live).  They must leave nothing
    behind: identical segmentation / embeddings / clustering / speech turns with warmup=0, stream clocks
    at zero, lanes starting at 0."""
    n, W, hop, steps = 4, 80000, 8000, 6
    audio = torch.from_numpy(synth_streams(n, (W + hop * steps) / 16000.0, seed0=77)).to(gpu)
    seg_sd, emb_sd = synth_segmentation_state(), synth_embedding_state()

    def run(warm):
        sb = StreamBatch(M.HipSegmentation(seg_sd, max_batch=n), M.HipEmbedding(emb_sd, max_batch=n), n,
                         device=gpu, tail=True, latency=1.0, warmup=warm)
        out = []
        for t in range(steps):
            ticket = sb.launch(audio[:, t * hop:t * hop + W])
            if t == 0:
                assert sb._real_steps == 0 and sb._t == 1 and (sb._steps == 1).all() and sb.tail is not None
            res = [np.array(x) for x in sb.finish(ticket)]
            agg, rows, t0, r, turns, nturns = ticket["tail"]
            out.append(res + [np.array(rows), np.array(t0), np.array(nturns),
                              np.concatenate([turns[i, :nturns[i]].ravel() for i in range(n)])])
        assert sb._real_steps == steps
        return out

    cold, warm = run(0), run(10)
    for a, b in zip(cold, warm):
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)
