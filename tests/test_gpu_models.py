"""Whole-network parity on the GPU: the HIP path (through the C ABI and the operator API) vs the
CPU oracle (oracle/models_ref.py) on the same seeded weights and synthetic audio.

Tolerances (stated, north-star: "within a stated tolerance"; SURVEY.md 8c / BASELINE.md 4.6): segmentation
max|d| <= 1e-4, mean|d| <= 3e-5 (observed ~9e-6 / ~9e-7: the HIP path is another fp32 summation
order through 4 BiLSTM layers x 293 steps); embeddings cosine >= 0.99999 and relative L2 <= 1e-4.
"""
import numpy as np
import pytest
import torch

from diart_amd import _lib
from diart_amd import models as M
from diart_amd.synth import (sliding_chunks, synth_embedding_state, synth_segmentation_state,
                             synth_stream)

pytestmark = pytest.mark.gpu

SEG_MAX, SEG_MEAN, EMB_COS = 1e-4, 3e-5, 0.99999     # seg max|d| <= 1e-4


@pytest.fixture(scope="module")
def chunks():
    st = synth_stream(11, 12.0)
    return torch.from_numpy(sliding_chunks(st).copy())[:, None, :]  # (15,1,80000)


@pytest.fixture(scope="module")
def oracle_models():
    from oracle.models_ref import PyanNetRef, XVectorSincNetRef
    s, e = PyanNetRef().eval(), XVectorSincNetRef().eval()
    s.load_state_dict(synth_segmentation_state())
    e.load_state_dict(synth_embedding_state())
    return s, e


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_segmentation_forward(gpu, chunks, oracle_models, precision):
    """Both arithmetic modes against the SAME gate: "f16x3" (split-f16 products on the half-precision
    matrix cores for the LSTM projections and the MLP) is not given a looser tolerance."""
    seg = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=8, precision=precision)
    seg.to(gpu)
    x = chunks[:7]
    with torch.no_grad():
        ref = oracle_models[0](x)
    got = seg(x.to(gpu)).cpu()
    assert got.shape == ref.shape == (7, 293, 3)
    d = (got - ref).abs()
    print(precision, "seg max|d|", d.max().item(), "mean|d|", d.mean().item())
    assert d.max().item() < SEG_MAX and d.mean().item() < SEG_MEAN
    # batch invariance (README.md:430): B=1 and a strided rolling-window view give the same rows
    one = seg(x[3:4].to(gpu)).cpu()
    assert (one[0] - got[3]).abs().max().item() < 1e-6
    stream = torch.from_numpy(synth_stream(11, 12.0)).to(gpu)
    view = stream.unfold(0, 80000, 8000)[:7]  # rolling window addressed in place
    assert view.stride(0) == 8000
    got2 = seg(view[:, None, :]).cpu()
    assert (got2 - got).abs().max().item() < 1e-6


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_embedding_forward_both_forms(gpu, chunks, oracle_models, precision):
    emb = M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=16, precision=precision)
    emb.to(gpu)
    x = chunks[:4]
    g = torch.Generator().manual_seed(0)
    w = torch.rand(4, 293, 3, generator=g) ** 2 + 1e-8       # (B,F,K) like OSP output
    with torch.no_grad():
        ref = oracle_models[1].forward_multi(x, w)          # (B,K,512)
    # reference-style call: (B*K) repeated rows, "(batch spk) frame" weights (embedding.py:56-59)
    rows = x.repeat(1, 3, 1).reshape(12, 1, -1).to(gpu)
    wrows = w.permute(0, 2, 1).reshape(12, 293).to(gpu)
    got_rows = emb(rows, wrows).cpu().view(4, 3, 512)
    # de-duplicated call
    got_multi = emb.model.forward_multi(x.to(gpu), w.permute(0, 2, 1).contiguous().to(gpu)).cpu()
    for got in (got_rows, got_multi):
        cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=-1)
        rel = ((got - ref).norm(dim=-1) / ref.norm(dim=-1)).max().item()
        print(precision, "emb cos min", cos.min().item(), "rel", rel)
        assert cos.min().item() >= EMB_COS and rel < 1e-4
    assert (got_rows - got_multi).abs().max().item() < 1e-5
    # no weights -> plain mean / unbiased std pooling
    with torch.no_grad():
        ref0 = oracle_models[1](x)
    got0 = emb(x.to(gpu)).cpu()
    assert ((got0 - ref0).norm(dim=-1) / ref0.norm(dim=-1)).max().item() < 1e-4
    # normalised form
    gotn = emb.model.forward_multi(x.to(gpu), w.permute(0, 2, 1).contiguous().to(gpu), normalize=True).cpu()
    assert torch.allclose(gotn.norm(dim=-1), torch.ones(4, 3), atol=1e-5)


def test_powerset_segmentation(gpu, chunks):
    from oracle.models_ref import PyanNetRef, powerset_to_multilabel
    sd = synth_segmentation_state(seed=77, powerset=True)
    ref_m = PyanNetRef(powerset=True).eval()
    ref_m.load_state_dict(sd)
    seg = M.SegmentationModel.from_state(sd, max_batch=4, powerset=True).to(gpu)
    x = chunks[:3]
    with torch.no_grad():
        logp = ref_m(x)
        ref = powerset_to_multilabel(logp)
    got = seg(x.to(gpu)).cpu()
    assert got.shape == (3, 293, 3) and set(np.unique(got.numpy())) <= {0.0, 1.0}
    # hard decisions may flip only where the top-2 log-probs are within fp32 noise
    top2 = logp.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(got[safe], ref[safe])
    assert (got != ref).float().mean().item() < 0.01


def test_errors_are_loud(gpu):
    from diart_amd._lib import DiartAmdError
    seg = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=2).to(gpu)
    with pytest.raises(DiartAmdError):
        seg(torch.zeros(1, 1, 100, device=gpu))      # too short for SincNet
    with pytest.raises(ValueError):
        seg(torch.zeros(1, 2, 80000, device=gpu))    # not mono
    with pytest.raises(DiartAmdError):
        M.HipSegmentation(synth_segmentation_state()).to(torch.device("cpu"))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_other_geometries_10s_windows_4_speakers(gpu, precision):
    """Edge geometries the reference supports through its config (duration, model variant):
    10 s windows (160000 samples -> 589 frames, 575 x-vector frames), the 4-speaker multilabel head
    (`@Interspeech2021`), batch 1 and a batch that outgrows ``max_batch`` (the handle
    is re-created), a stream row that is only 16-byte aligned; both arithmetic modes."""
    from oracle.models_ref import PyanNetRef, XVectorSincNetRef
    seg_sd, emb_sd = synth_segmentation_state(seed=5, num_speakers=4), synth_embedding_state(seed=6)
    ref_s, ref_e = PyanNetRef(num_speakers=4).eval(), XVectorSincNetRef().eval()
    ref_s.load_state_dict(seg_sd)
    ref_e.load_state_dict(emb_sd)
    seg = M.HipSegmentation(seg_sd, max_batch=2, precision=precision).to(gpu)
    emb = M.HipEmbedding(emb_sd, max_batch=2, precision=precision).to(gpu)
    stream = torch.from_numpy(synth_stream(21, 13.0))
    assert seg.num_frames(160000) == 589
    for B in (1, 3):                                           # 3 > max_batch: handle grows
        x = torch.stack([stream[4 + 8000 * i: 4 + 8000 * i + 160000] for i in range(B)])[:, None, :]
        with torch.no_grad():
            rs = ref_s(x)
            w = torch.rand(B, 589, 4, generator=torch.Generator().manual_seed(B)) ** 2 + 1e-8
            re = ref_e.forward_multi(x, w)
        dx = stream.to(gpu)[4:].unfold(0, 160000, 8000)[:B]    # in-place view, offset 4 samples = 16 B
        assert dx.data_ptr() % 16 == 0 and dx.stride(0) == 8000
        gs = seg(dx[:, None, :]).cpu()
        assert gs.shape == rs.shape == (B, 589, 4)
        assert (gs - rs).abs().max().item() < SEG_MAX and (gs - rs).abs().mean().item() < SEG_MEAN
        ge = emb.forward_multi(dx[:, None, :], w.permute(0, 2, 1).contiguous().to(gpu)).cpu()
        assert ge.shape == re.shape == (B, 4, 512)
        assert ((ge - re).norm(dim=-1) / re.norm(dim=-1)).max().item() < 1e-4


def test_precisions_agree_with_each_other(gpu, chunks):
    """f16x3 vs f32 directly: closer to each other than either is allowed to be to the oracle."""
    x = chunks[:6].to(gpu)
    out = {}
    for p in ("f32", "f16x3"):
        seg = M.HipSegmentation(synth_segmentation_state(), max_batch=8, precision=p).to(gpu)
        emb = M.HipEmbedding(synth_embedding_state(), max_batch=8, precision=p).to(gpu)
        s = seg(x)
        from diart_amd.functional import overlapped_speech_penalty
        w = overlapped_speech_penalty(s, 3, 10, speaker_major=True)
        out[p] = (s.cpu(), emb.forward_multi(x, w, normalize=True).cpu())
    ds = (out["f32"][0] - out["f16x3"][0]).abs().max().item()
    de = (out["f32"][1] - out["f16x3"][1]).abs().max().item()
    print("f32 vs f16x3: seg max|d|", ds, "normalised emb max|d|", de)
    assert ds < 5e-5 and de < 5e-6


@pytest.mark.parametrize("B", [2, 12])
def test_nearest_weight_interpolation_of_pyannote_3_1(gpu, oracle_models, B):
    """pyannote.audio >= 3.1 resamples StatsPool's weights with mode="nearest" (2.x .. 3.0: "linear"; the reference pins
    >= 2.1.1, /root/reference/setup.cfg:35).  `weight_interp="nearest"` against the oracle's `interp_mode="nearest"`,
    through the stand-alone pooling kernel (2 chunks: the latency regime) and the pooled tdnn5 epilogue (12 chunks), the
    de-duplicated and the reference-shaped call; and it is NOT the linear answer."""
    from diart_amd.synth import synth_streams
    x = torch.from_numpy(synth_streams(B, 5.0, seed0=60))[:, None, :80000].contiguous()
    g = torch.Generator().manual_seed(5)
    w = torch.rand(B, 3, 293, generator=g) ** 4 + 1e-8              # speaker-major (B, K, F): peaky, so the mode matters
    ref_model = oracle_models[1]
    outs = {}
    for interp in ("nearest", "linear"):
        emb = M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=3 * B, weight_interp=interp)
        emb.to(gpu)
        assert emb.model.weight_interp == interp
        multi = emb.model.forward_multi(x.to(gpu), w.to(gpu)).cpu()
        rows = emb(x.repeat(1, 3, 1).reshape(3 * B, 1, -1).to(gpu), w.reshape(3 * B, 293).to(gpu)).cpu().view(B, 3, 512)
        outs[interp] = (multi, rows)
    with torch.no_grad():
        refs = {m: ref_model.forward_multi(x, w.permute(0, 2, 1), interp_mode=m) for m in ("nearest", "linear")}
    for interp in ("nearest", "linear"):
        for got in outs[interp]:
            rel = ((got - refs[interp]).norm(dim=-1) / refs[interp].norm(dim=-1)).max().item()
            assert rel < 1e-4, (interp, rel)
    assert ((refs["nearest"] - refs["linear"]).norm(dim=-1) / refs["linear"].norm(dim=-1)).max().item() > 1e-3
    assert ((outs["nearest"][0] - refs["linear"]).norm(dim=-1) / refs["linear"].norm(dim=-1)).max().item() > 1e-3


def test_pooling_fused_into_tdnn5_equals_the_two_launch_path(gpu, oracle_models, monkeypatch):
    """Round 3: tdnn5 keeps its 128 x 128 output tile in LDS and reduces it to weighted moments there
    (k_gemm_pre.hip pooled epilogue + pool_combine) instead of writing 110 MB of frame features for
    stats_pool to read back.  Same embeddings as the two-launch path (option pool_fuse = 0) and as the
    oracle, for the de-duplicated call (K = 3 speakers per chunk), the reference-shaped rows call
    (K = 1) and the unweighted call; 12 chunks so that the launch uses big tiles (the latency regime
    keeps the two-launch path)."""
    from diart_amd.synth import synth_streams
    B = 12
    x = torch.from_numpy(synth_streams(B, 5.0, seed0=40))[:, None, :80000].contiguous()
    g = torch.Generator().manual_seed(3)
    w = torch.rand(B, 3, 293, generator=g) ** 2 + 1e-8              # speaker-major (B, K, F)
    w[1, 2] = 1e-8                                                  # a silent speaker
    w[2, :, 100:] = 1e-8                                            # weight only in the first pieces
    outs = {}
    for fuse in ("1", "0"):
        _lib.set_option("pool_fuse", int(fuse))
        try:
            emb = M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=3 * B)
            emb.to(gpu)
            multi = emb.model.forward_multi(x.to(gpu), w.to(gpu)).cpu()
            rows = emb(x.repeat(1, 3, 1).reshape(3 * B, 1, -1).to(gpu), w.reshape(3 * B, 293).to(gpu)).cpu().view(B, 3, 512)
            plain = emb(x.to(gpu)).cpu()
        finally:
            _lib.set_option("pool_fuse", 1)
        outs[fuse] = (multi, rows, plain)
    with torch.no_grad():
        ref = oracle_models[1].forward_multi(x, w.permute(0, 2, 1))
        ref0 = oracle_models[1](x)
    for a, b_, name in zip(outs["1"], outs["0"], ("multi", "rows", "plain")):
        rel = ((a - b_).norm(dim=-1) / b_.norm(dim=-1)).max().item()
        print("fused vs two-launch", name, rel)
        assert rel < 2e-6, name
    for got in outs["1"][:2]:
        cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=-1)
        assert cos.min().item() >= EMB_COS and ((got - ref).norm(dim=-1) / ref.norm(dim=-1)).max().item() < 1e-4
    assert ((outs["1"][2] - ref0).norm(dim=-1) / ref0.norm(dim=-1)).max().item() < 1e-4


def _degenerate_windows():
    """Windows a live service and the reference's file reader really produce: digital silence, a constant (DC) level,
    a file's last chunk padded with zeros (/root/reference/src/diart/sources.py:117-120), a nearly silent recording,
    full-scale clipping, one click, a window whose loud part is short."""
    g = torch.Generator().manual_seed(123)
    S = 80000
    speech = torch.from_numpy(synth_stream(17, 5.0))[:S]
    rows = {
        "silence": torch.zeros(S),
        "dc": torch.full((S,), 0.25),
        "zero_padded_tail": torch.cat([speech[:30000], torch.zeros(S - 30000)]),
        "nearly_silent": 1e-6 * torch.randn(S, generator=g),
        "clipped_square": torch.sign(torch.sin(torch.arange(S) * 2 * np.pi * 220 / 16000)),
        "click": torch.zeros(S).index_fill_(0, torch.tensor([40000]), 1.0),
        "loud_burst_in_silence": torch.cat([torch.zeros(60000), 0.9 * torch.randn(2000, generator=g).clamp(-1, 1), torch.zeros(18000)]),
        "speech": speech,
    }
    return list(rows), torch.stack(list(rows.values()))[:, None, :].float()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_degenerate_windows_match_the_oracle(gpu, oracle_models, precision):
    """Same gates as ordinary audio, both arithmetic modes; no NaN / Inf and no range flag (a window of silence
    normalises to zeros, a constant to zeros too: InstanceNorm's eps keeps both finite)."""
    names, x = _degenerate_windows()
    seg = M.HipSegmentation(synth_segmentation_state(), max_batch=len(names), precision=precision).to(gpu)
    emb = M.HipEmbedding(synth_embedding_state(), max_batch=len(names), precision=precision).to(gpu)
    w = torch.rand(len(names), 293, 3, generator=torch.Generator().manual_seed(5)) ** 2 + 1e-8
    with torch.no_grad():
        rs = oracle_models[0](x)
        re = oracle_models[1].forward_multi(x, w)
        re0 = oracle_models[1](x)
    gs = seg(x.to(gpu)).cpu()
    ge = emb.forward_multi(x.to(gpu), w.permute(0, 2, 1).contiguous().to(gpu)).cpu()
    ge0 = emb(x.to(gpu)).cpu()
    _lib.range_check(gpu.index)
    assert torch.isfinite(gs).all() and torch.isfinite(ge).all() and torch.isfinite(ge0).all()
    for i, nm in enumerate(names):
        d = (gs[i] - rs[i]).abs()
        rel = ((ge[i] - re[i]).norm(dim=-1) / re[i].norm(dim=-1)).max().item()
        rel0 = ((ge0[i] - re0[i]).norm() / re0[i].norm()).item()
        cos = torch.nn.functional.cosine_similarity(ge[i].double(), re[i].double(), dim=-1).min().item()
        print(f"{precision} {nm}: seg max|d| {d.max().item():.2e}, emb rel {rel:.2e} / unweighted {rel0:.2e}, cos {cos:.7f}")
        assert d.max().item() < SEG_MAX and d.mean().item() < SEG_MEAN, nm
        assert cos >= EMB_COS and rel < 1e-4 and rel0 < 1e-4, nm


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_nan_or_inf_samples_make_nan_rows_like_torch(gpu, chunks, oracle_models, precision):
    """One NaN (or Inf) sample in a window: PyTorch's InstanceNorm1d makes the whole window NaN, so the reference's
    segmentation and embeddings of that row are NaN and its clustering drops the chunk's speakers
    (/root/reference/src/diart/blocks/clustering.py:137-145).  The exact-f32 kernels propagate NaN by themselves; the
    split-f16 kernels clamp operands (NaN becomes a finite value there), so their last kernels write the NaN rows from
    the waveform statistics.  Every other row of the batch is bit-identical to the clean batch, through the model calls
    and through a StreamBatch step (whose stream with the bad window gets no speaker for it)."""
    from diart_amd.pipeline import StreamBatch
    x = chunks[:5].clone()
    clean = x.clone()
    x[1, 0, 40000] = float("nan")
    x[3, 0, 7] = float("inf")
    with torch.no_grad():
        rs, re = oracle_models[0](x), oracle_models[1](x)
    assert torch.isnan(rs[[1, 3]]).all() and torch.isnan(re[[1, 3]]).all() and torch.isfinite(rs[[0, 2, 4]]).all()
    seg = M.HipSegmentation(synth_segmentation_state(), max_batch=8, precision=precision).to(gpu)
    emb = M.HipEmbedding(synth_embedding_state(), max_batch=8, precision=precision).to(gpu)
    w = (torch.rand(5, 3, 293, generator=torch.Generator().manual_seed(2)) ** 2 + 1e-8).to(gpu)
    gs, gs0 = seg(x.to(gpu)).cpu(), seg(clean.to(gpu)).cpu()
    ge, ge0 = emb(x.to(gpu)).cpu(), emb(clean.to(gpu)).cpu()
    gm, gm0 = emb.forward_multi(x.to(gpu), w).cpu(), emb.forward_multi(clean.to(gpu), w).cpu()
    rows = x.repeat(1, 3, 1).reshape(15, 1, -1).to(gpu)                 # the reference's (batch spk) rows
    gr = emb(rows, w.reshape(15, 293)).cpu().view(5, 3, 512)
    for bad in (1, 3):
        assert torch.isnan(gs[bad]).all() and torch.isnan(ge[bad]).all() and torch.isnan(gm[bad]).all() and torch.isnan(gr[bad]).all()
    good = [0, 2, 4]
    assert torch.equal(gs[good], gs0[good]) and torch.equal(ge[good], ge0[good]) and torch.equal(gm[good], gm0[good])
    assert torch.isfinite(gr[good]).all()
    _lib.range_check(gpu.index)                                         # not an out-of-range event
    # one StreamBatch step: 5 streams, the same windows
    pipe = StreamBatch(M.HipSegmentation(synth_segmentation_state(), max_batch=5, precision=precision),
                       M.HipEmbedding(synth_embedding_state(), max_batch=5, precision=precision), 5, device=gpu)
    s1, e1, scores, assign = pipe(x[:, 0, :].to(gpu))
    assert np.isnan(s1[[1, 3]]).all() and np.isnan(e1[[1, 3]]).all() and np.isfinite(s1[good]).all()
    assert (assign[[1, 3]] < 0).all(), assign                            # no speaker of the bad windows is assigned
    assert np.isfinite(scores[good]).all()
