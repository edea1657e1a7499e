"""RTTM / DER parity of the whole pipeline (BASELINE.json config 1 shape: one 30 s 16 kHz stream,
5 s window, 500 ms step, Benchmark's batch of 32): ``diart_amd.blocks.SpeakerDiarization`` on the
GPU vs the all-CPU chain (oracle networks -> oracle clustering -> oracle aggregation/binarise).

Gate (north-star: "DER within 0.5 pt of reference"): DER of the GPU hypothesis scored against the
CPU-chain hypothesis <= 0.5 %, for latency = 0.5 s and 5 s; batch_size 1 and 32 give identical
RTTMs (README.md:430).  Same for the VoiceActivityDetection pipeline (config 5, step 0.25 s).
"""
import numpy as np
import pytest
import torch

from diart_amd import models as M
from diart_amd.blocks import (SpeakerDiarization, SpeakerDiarizationConfig, VoiceActivityDetection,
                              VoiceActivityDetectionConfig)
from diart_amd.features import Annotation, Segment, SlidingWindow, SlidingWindowFeature
from diart_amd.metrics import DetectionErrorRate, DiarizationErrorRate
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream

pytestmark = pytest.mark.gpu
SR = 16000


def rolling_chunks(stream: np.ndarray, duration=5.0, step=0.5):
    """What rearrange_audio_stream emits (operators.py:44-100): (samples, 1) windows with timestamps."""
    S, H = int(round(duration * SR)), int(round(step * SR))
    out = []
    for i in range((len(stream) - S) // H + 1):
        sw = SlidingWindow(start=i * step, duration=1 / SR, step=1 / SR)
        out.append(SlidingWindowFeature(stream[i * H:i * H + S, None], sw))
    return out


def accumulate(outputs, uri="stream") -> Annotation:
    """PredictionAccumulator (sinks.py:59-88): update() then support(0.05)."""
    total = Annotation(uri)
    for ann, _ in outputs:
        total.update(ann)
    return total.support(0.05)


def cpu_chain(stream, latency, tau=0.6, rho=0.3, delta=1.0, step=0.5):
    from oracle.clustering_ref import OnlineSpeakerClusteringRef
    from oracle.functional_ref import normalize_embeddings_ref, overlapped_speech_penalty_ref
    from oracle.models_ref import PyanNetRef, XVectorSincNetRef
    from oracle.pyannote_stub import SlidingWindow as SW, SlidingWindowFeature as SWF
    from oracle.tail_ref import TailRef
    seg_m, emb_m = PyanNetRef().eval(), XVectorSincNetRef().eval()
    seg_m.load_state_dict(synth_segmentation_state())
    emb_m.load_state_dict(synth_embedding_state())
    clu, tail = OnlineSpeakerClusteringRef(tau, rho, delta, "cosine", 20), TailRef(step, latency, tau)
    total = Annotation("stream")
    chunks = rolling_chunks(stream, 5.0, step)
    x = torch.from_numpy(np.stack([c.data[:, 0] for c in chunks]))[:, None, :]
    with torch.no_grad():
        seg = torch.cat([seg_m(x[i:i + 8]) for i in range(0, len(chunks), 8)])
        emb = torch.cat([normalize_embeddings_ref(emb_m.forward_multi(x[i:i + 8], overlapped_speech_penalty_ref(seg[i:i + 8])))
                         for i in range(0, len(chunks), 8)])
    for i, c in enumerate(chunks):
        scores, _ = clu(seg[i].numpy(), emb[i].numpy())
        res = 5.0 / seg.shape[1]
        _, turns = tail(SWF(scores, SW(start=i * step, duration=res, step=res)))
        for n, (s, e, spk) in enumerate(turns):
            total[Segment(s, e), (i, n)] = f"speaker{spk}"
    return total.support(0.05), seg.numpy()


@pytest.fixture(scope="module")
def stream():
    return synth_stream(2024, 30.0)


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def models(request):
    """Every RTTM / DER gate below runs in both arithmetic modes of the GEMM-shaped layers: the
    default split-f16 MFMA path and the exact-f32 MFMA path (weights.PRECISIONS)."""
    p = request.param
    return (M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=32, precision=p),
            M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=32, precision=p))


@pytest.mark.parametrize("latency", [0.5, 5.0])
def test_diarization_rttm_matches_cpu_chain(gpu, stream, models, latency):
    cfg = SpeakerDiarizationConfig(segmentation=models[0], embedding=models[1], latency=latency, device=gpu)
    chunks = rolling_chunks(stream)
    assert len(chunks) == 51                                       # ceil((30-5+0.5)/0.5), inference.py:81-83
    hyps = {}
    for bs in (32, 1):
        pipe = SpeakerDiarization(cfg)
        outs = []
        for i in range(0, len(chunks), bs):
            outs += pipe(chunks[i:i + bs])
        assert len(outs) == len(chunks)
        hyps[bs] = accumulate(outs)
        # audio aggregation returns the `step` seconds of audio that the prediction is about
        assert all(w.data.shape[1] == 1 for _, w in outs)
    assert hyps[32].to_rttm() == hyps[1].to_rttm(), "batch size changed the output"
    ref, seg = cpu_chain(stream, latency)
    assert len(ref) > 0 and seg.max() > 0.6, "degenerate scenario: nobody ever speaks"
    der = DiarizationErrorRate()
    d = der(ref, hyps[32], detailed=True)
    print(f"latency {latency}: DER(GPU vs CPU chain) = {100 * d['diarization error rate']:.3f} % "
          f"of {d['total']:.1f} s; turns {len(hyps[32])} vs {len(ref)}")
    assert d["diarization error rate"] <= 0.005


def test_vad_matches_cpu_chain(gpu, stream, models):
    from oracle.models_ref import PyanNetRef
    from oracle.pyannote_stub import SlidingWindow as SW, SlidingWindowFeature as SWF
    from oracle.tail_ref import TailRef
    step = 0.25
    cfg = VoiceActivityDetectionConfig(segmentation=models[0], step=step, latency=step, device=gpu)
    pipe = VoiceActivityDetection(cfg)
    chunks = rolling_chunks(stream, 5.0, step)
    outs = []
    for i in range(0, len(chunks), 16):
        outs += pipe(chunks[i:i + 16])
    hyp = accumulate(outs)
    assert set(hyp.labels()) <= {"speech"}
    seg_m = PyanNetRef().eval()
    seg_m.load_state_dict(synth_segmentation_state())
    tail, ref = TailRef(step, step, 0.6), Annotation("stream")
    for i, c in enumerate(chunks):
        with torch.no_grad():
            s = seg_m(torch.from_numpy(c.data[:, 0])[None, None, :])[0].max(dim=-1, keepdim=True)[0].numpy()
        _, turns = tail(SWF(s, SW(start=i * step, duration=5.0 / 293, step=5.0 / 293)))
        for n, (a, b, _) in enumerate(turns):
            ref[Segment(a, b), (i, n)] = "speech"
    d = DetectionErrorRate()(ref.support(0.05), hyp, detailed=True)
    print(f"VAD: DetER(GPU vs CPU chain) = {100 * d['detection error rate']:.3f} % of {d['total']:.1f} s")
    assert d["total"] > 1.0 and d["detection error rate"] <= 0.005


def test_pipeline_asserts_like_the_reference(gpu, models):
    cfg = SpeakerDiarizationConfig(segmentation=models[0], embedding=models[1], device=gpu)
    pipe = SpeakerDiarization(cfg)
    with pytest.raises(AssertionError):
        pipe([])
    bad = SlidingWindowFeature(np.zeros((100, 1), dtype=np.float32), SlidingWindow(start=0, duration=1 / SR, step=1 / SR))
    with pytest.raises(AssertionError):
        pipe([bad])
    with pytest.raises(AssertionError):
        SpeakerDiarization(SpeakerDiarizationConfig(segmentation=models[0], embedding=models[1], latency=7, device=gpu))
    assert [h.name for h in SpeakerDiarization.hyper_parameters()] == ["tau_active", "rho_update", "delta_new"]
    assert SpeakerDiarization.get_config_class() is SpeakerDiarizationConfig


def test_config3_powerset_ecapa_pipeline_matches_cpu_chain(gpu):
    """BASELINE.json config 3: segmentation-3.0 (powerset -> hard multilabel) + ECAPA-TDNN with
    normalised OSP weights, 12 s stream, Benchmark-style batches; RTTM vs the all-CPU chain."""
    from oracle.clustering_ref import OnlineSpeakerClusteringRef
    from oracle.ecapa_ref import PretrainedSpeakerEmbeddingRef
    from oracle.functional_ref import normalize_embeddings_ref, overlapped_speech_penalty_ref
    from oracle.models_ref import PyanNetRef, powerset_to_multilabel
    from oracle.pyannote_stub import SlidingWindow as SW, SlidingWindowFeature as SWF
    from oracle.tail_ref import TailRef
    from diart_amd.synth import synth_ecapa_state
    stream = synth_stream(31, 12.0)
    seg_sd, emb_sd = synth_segmentation_state(seed=77, powerset=True), synth_ecapa_state()
    cfg = SpeakerDiarizationConfig(
        segmentation=M.SegmentationModel.from_state(seg_sd, max_batch=16, powerset=True),
        embedding=M.EmbeddingModel.from_state(emb_sd, max_batch=48), latency=0.5, tau_active=0.5,
        normalize_embedding_weights=True, device=gpu)
    pipe = SpeakerDiarization(cfg)
    chunks = rolling_chunks(stream)
    outs = []
    for i in range(0, len(chunks), 8):
        outs += pipe(chunks[i:i + 8])
    hyp = accumulate(outs)
    # ---- all-CPU chain ------------------------------------------------------------------
    seg_m = PyanNetRef(powerset=True).eval()
    seg_m.load_state_dict(seg_sd)
    emb_m = PretrainedSpeakerEmbeddingRef(emb_sd)
    clu, tail, ref = OnlineSpeakerClusteringRef(0.5, 0.3, 1.0, "cosine", 20), TailRef(0.5, 0.5, 0.5), Annotation("stream")
    # a second, FULLY independent CPU chain (its own hard segmentation from the CPU network): the
    # end-to-end GPU-vs-CPU comparison, gated with a budget for the near-tie flips counted below
    clu_i, tail_i, ref_i = OnlineSpeakerClusteringRef(0.5, 0.3, 1.0, "cosine", 20), TailRef(0.5, 0.5, 0.5), Annotation("stream")
    # Hard powerset decisions flip where the two best classes are within fp32 noise of each other
    # (random weights produce many such near-ties).  Those flips are checked on their own — they
    # may only happen at near-ties — and the rest of the chain (OSP -> masks -> ECAPA ->
    # clustering -> aggregation) is then compared on the SAME hard segmentation.
    flips = near_ties = 0
    for i0 in range(0, len(chunks), 8):
        # the reference hands the embedding model all (chunk, speaker) rows of a batch at once, and
        # ECAPA's padding makes a row depend on the longest row of its call: batch like the GPU run
        batch = chunks[i0:i0 + 8]
        x = torch.from_numpy(np.stack([c.data[:, 0] for c in batch]))[:, None, :]
        with torch.no_grad():
            logp = seg_m(x)
        cpu_seg = powerset_to_multilabel(logp)                                      # (B,293,3) in {0,1}
        seg = cfg.segmentation(x.to(gpu)).cpu()
        top2 = logp.topk(2, dim=-1).values
        margin = top2[..., 0] - top2[..., 1]
        differ = (seg != cpu_seg).any(dim=-1)
        assert (margin[differ] < 1e-3).all(), "a hard decision flipped away from a near-tie"
        flips += int(differ.sum())
        near_ties += int((margin < 1e-3).sum())
        B = len(batch)
        rows = x.repeat(1, 3, 1).reshape(B * 3, 1, -1)
        for which_seg, c_, t_, r_ in ((seg, clu, tail, ref), (cpu_seg, clu_i, tail_i, ref_i)):
            w = overlapped_speech_penalty_ref(which_seg)
            mn, mx = w.min(dim=1, keepdim=True).values, w.max(dim=1, keepdim=True).values
            w = ((w - mn) / (mx - mn)).nan_to_num(1e-8)
            emb = torch.from_numpy(emb_m(rows, w.permute(0, 2, 1).reshape(B * 3, -1))).view(B, 3, -1)
            emb = normalize_embeddings_ref(emb)
            for j in range(B):
                i = i0 + j
                scores, _ = c_(which_seg[j].numpy(), emb[j].numpy())
                _, turns = t_(SWF(scores, SW(start=i * 0.5, duration=5 / 293, step=5 / 293)))
                for n, (a, b, spk) in enumerate(turns):
                    r_[Segment(a, b), (i, n)] = f"speaker{spk}"
    ref, ref_i = ref.support(0.05), ref_i.support(0.05)
    d = DiarizationErrorRate()(ref, hyp, detailed=True)
    di = DiarizationErrorRate()(ref_i, hyp, detailed=True)
    # budget of the independent comparison: the north-star 0.5 pt plus, for every flipped hard
    # decision, the frame it changes in the (latency = step: one window per region) output and the
    # two neighbours a changed mask can move through the ECAPA embedding -> 3 frames of 5/293 s
    budget = 0.005 + 3 * flips * (5 / 293) / max(di["total"], 1e-9)
    print(f"config 3: DER(GPU vs CPU chain fed the GPU's hard segmentation) = {100 * d['diarization error rate']:.3f} % "
          f"of {d['total']:.1f} s; DER(GPU vs fully independent CPU chain) = "
          f"{100 * di['diarization error rate']:.3f} % (budget {100 * budget:.3f} %); "
          f"{flips} hard-decision flips at {near_ties} near-tie frames of {293 * len(chunks)}")
    assert d["total"] > 1.0 and d["diarization error rate"] <= 0.005
    assert di["total"] > 1.0 and di["diarization error rate"] <= budget


def test_benchmark_over_wav_files_matches_cpu_chain(gpu, models, tmp_path):
    """BASELINE.json config 1 as the reference runs it: WAV files in a directory ->
    ``Benchmark(speech, reference, output, batch_size=32)(SpeakerDiarization, config)`` ->
    one RTTM per file + the DER report.  The reference RTTMs here are the all-CPU chain's
    hypotheses on the same (16-bit) audio, so the reported DER IS the GPU-vs-CPU parity figure
    (north-star gate: within 0.5 pt)."""
    from diart_amd.inference import Benchmark, DistributedBenchmark, read_wav, write_wav
    speech, refs, out = tmp_path / "wav", tmp_path / "ref", tmp_path / "out"
    speech.mkdir()
    refs.mkdir()
    for name, seed, dur in (("meeting_a", 2024, 30.0), ("meeting_b", 7, 12.0)):
        write_wav(speech / f"{name}.wav", synth_stream(seed, dur), SR)
        audio, sr = read_wav(speech / f"{name}.wav")
        assert sr == SR
        ref, _ = cpu_chain(audio, 0.5)
        ref.uri = name
        with open(refs / f"{name}.rttm", "w") as f:
            ref.write_rttm(f)
    cfg = SpeakerDiarizationConfig(segmentation=models[0], embedding=models[1], latency=0.5, device=gpu)
    metric = Benchmark(speech, refs, out, show_report=False, batch_size=32)(SpeakerDiarization, cfg)
    print(metric.report())
    assert [u for u, _ in metric.results] == ["meeting_a", "meeting_b"]
    assert metric.accumulated["total"] > 5.0 and abs(metric) <= 0.005
    assert sorted(p.name for p in out.iterdir()) == ["meeting_a.rttm", "meeting_b.rttm"]
    # single-rank DistributedBenchmark is the same run
    m2 = DistributedBenchmark(Benchmark(speech, refs, tmp_path / "out2", show_report=False))(SpeakerDiarization, cfg)
    assert m2.accumulated == metric.accumulated
    assert (tmp_path / "out2" / "meeting_a.rttm").read_text() == (out / "meeting_a.rttm").read_text()


@pytest.mark.parametrize("latency", [0.5, 2.0])
def test_files_batched_together_give_the_rttm_files_of_the_one_file_at_a_time_loop(gpu, tmp_path, latency):
    """``Benchmark`` runs the files of a rank concurrently through ``FileBatch``
    (consecutive windows of several files per GPU step, the C++ clustering + output tail walking each
    file in order) — the RTTM files must be BYTE-identical to the reference-shaped loop (one file at a
    time, batches of 32 windows, per-chunk Python tail), whatever the number of concurrent files; file
    lengths include one that ends mid-block, one shorter than a window (left padding) and one with
    fewer windows than the others have per step."""
    from diart_amd.inference import Benchmark, write_wav
    speech = tmp_path / "wav"
    speech.mkdir()
    for name, seed, dur in (("a", 11, 21.3), ("b", 12, 9.05), ("c", 13, 33.0), ("d", 14, 3.2), ("e", 15, 5.5),
                            ("f", 16, 14.77)):
        write_wav(speech / f"{name}.wav", synth_stream(seed, dur), SR)
    seg = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=64)
    emb = M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=64)
    cfg = SpeakerDiarizationConfig(segmentation=seg, embedding=emb, latency=latency, device=gpu)
    outs = {}
    for tag, k in (("loop", 0), ("k1", 1), ("k4", 4), ("k16", 16)):
        b = Benchmark(speech, None, tmp_path / tag, show_report=False, batch_size=32, concurrent_files=k)
        b(SpeakerDiarization, cfg)
        assert b.last_path == ("one_file_at_a_time" if k == 0 else "file_batch")
        outs[tag] = {p.name: p.read_text() for p in sorted((tmp_path / tag).iterdir())}
    assert sorted(outs["loop"]) == [f"{n}.rttm" for n in "abcdef"]
    assert any(len(t) > 0 for t in outs["loop"].values())
    for tag in ("k1", "k4", "k16"):
        assert outs[tag] == outs["loop"], f"{tag}: RTTM files differ from the one-file-at-a-time loop"
