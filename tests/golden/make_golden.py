"""Generate the golden fixtures from the REFERENCE'S OWN CODE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--out DIR]

Runs /root/reference/src/diart/{functional.py, mapping.py, blocks/clustering.py,
blocks/embedding.py} (loaded by path with the pyannote.core stand-in of oracle/pyannote_stub.py)
on the seeded inputs of scenarios.py and stores inputs + outputs as small .npz files next to this
script.  The fixtures pin oracle/functional_ref.py and oracle/clustering_ref.py, and through them
the HIP / C++ product path.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
# --out DIR: write somewhere else (tests/test_oracle_golden.py regenerates into a scratch directory
# and compares with the committed fixtures whenever /root/reference is present)
OUT = Path(sys.argv[sys.argv.index("--out") + 1]) if "--out" in sys.argv else HERE
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE))

from oracle.pyannote_stub import SlidingWindow, SlidingWindowFeature, load_reference  # noqa: E402
import scenarios  # noqa: E402


def main():
    ref = load_reference()
    # ---- functional.py -------------------------------------------------------------
    seg, emb = scenarios.functional_inputs()
    out = {}
    for gamma, beta in ((3, 10), (2, 5), (1.5, 10)):
        w = ref.functional.overlapped_speech_penalty(torch.from_numpy(seg), gamma, beta)
        out[f"osp_g{gamma}_b{beta}"] = w.numpy()
    for norm in (False, True):   # blocks/embedding.py:98-107
        block = ref.embedding.OverlappedSpeechPenalty(3, 10, normalize=norm)
        out[f"osp_block_norm{int(norm)}"] = block(torch.from_numpy(seg)).numpy()
    out["normalize"] = ref.functional.normalize_embeddings(torch.from_numpy(emb)).numpy()
    out["normalize_2d"] = ref.functional.normalize_embeddings(torch.from_numpy(emb[0]), norm=2.5).numpy()
    np.savez_compressed(OUT / "functional.npz", seg=seg, emb=emb, **out)

    # ---- blocks/embedding.py plumbing with a toy model (ordering / squeeze semantics) --------
    class Toy:  # custom-model contract of README.md:186-209: __call__ + .to(device)
        def to(self, device):
            return self

        def __call__(self, wave, weights=None):
            if weights is None:
                return torch.stack([wave[:, 0, :100].sum(-1), wave[:, 0, -100:].sum(-1)], -1)
            return torch.stack([wave[:, 0, :100].sum(-1), weights.sum(-1), weights[:, :10].sum(-1),
                                (weights * torch.arange(weights.shape[1])).sum(-1)], -1)
    toy = Toy()
    rng = np.random.default_rng(5)
    wav = rng.standard_normal((3, 800, 1)).astype(np.float32)
    sg = rng.random((3, 29, 3)).astype(np.float32)
    plumb = {"wav": wav, "sg": sg}
    model = ref.models.EmbeddingModel(lambda: toy)  # noqa: E731
    oase = ref.embedding.OverlapAwareSpeakerEmbedding(model, 3, 10, norm=1, device=torch.device("cpu"))
    plumb["oase_b3"] = oase(torch.from_numpy(wav), torch.from_numpy(sg)).numpy()
    plumb["oase_b1"] = oase(torch.from_numpy(wav[:1]), torch.from_numpy(sg[:1])).numpy()
    se = ref.embedding.SpeakerEmbedding(model, torch.device("cpu"))
    plumb["se_noweights"] = se(torch.from_numpy(wav)).numpy()
    np.savez_compressed(OUT / "embedding_plumbing.npz", **plumb)

    # ---- blocks/clustering.py ---------------------------------------------------------
    for name in scenarios.CLUSTERING:
        inp = scenarios.clustering_inputs(name)
        clu = ref.clustering.OnlineSpeakerClustering(inp["tau"], inp["rho"], inp["delta"], "cosine", inp["G"])
        T, F, K = inp["seg"].shape
        assign = -np.ones((T, K), dtype=np.int64)
        active = np.zeros((T, inp["G"]), dtype=np.int8)
        score_sum = np.zeros((T, inp["G"]))
        raised = np.zeros(T, dtype=np.int8)
        centers_trace = np.zeros((T, inp["G"]))
        for t in range(T):
            swf = SlidingWindowFeature(inp["seg"][t], SlidingWindow(start=0.5 * t, duration=5 / F, step=5 / F))
            try:
                res = clu(swf, torch.from_numpy(inp["emb"][t]))
            except (AssertionError, ValueError) as exc:
                raised[t] = 1
                print(f"  [{name}] step {t}: reference raised {type(exc).__name__}: {exc}")
                continue
            for g in range(inp["G"]):
                col = res.data[:, g]
                if col.any():
                    src = [k for k in range(K) if np.array_equal(col, inp["seg"][t][:, k].astype(np.float64))]
                    for k in src:
                        if assign[t, k] < 0:
                            assign[t, k] = g
                            break
            score_sum[t] = res.data.sum(0)
            active[t, sorted(clu.active_centers)] = 1
            centers_trace[t] = clu.centers.sum(1)
        np.savez_compressed(OUT / f"clustering_{name}.npz", seg=inp["seg"], emb=inp["emb"],
                            params=np.array([inp["tau"], inp["rho"], inp["delta"], inp["G"]]),
                            assign=assign, active=active, score_sum=score_sum, raised=raised,
                            centers_trace=centers_trace, centers=clu.centers)
        print(name, "steps", T, "final speakers", int(active[-1].sum()), "raised", int(raised.sum()))


def tail(ref):
    """blocks/aggregation.py DelayedAggregation + blocks/utils.py Binarize, driven by the buffer
    logic of blocks/diarization.py:203-232, for four latencies and two stream origins."""
    out = {}
    for start_time in (0.0, 7.5):
        scores, starts, res = scenarios.tail_inputs(start_time)
        tag0 = f"s{start_time:g}"
        for latency in scenarios.TAIL_LATENCIES:
            pred = ref.aggregation.DelayedAggregation(scenarios.TAIL_STEP, latency, strategy="hamming",
                                                      cropping_mode="loose")
            audio = ref.aggregation.DelayedAggregation(scenarios.TAIL_STEP, latency, strategy="first",
                                                       cropping_mode="center")
            mean = ref.aggregation.DelayedAggregation(scenarios.TAIL_STEP, latency, strategy="mean",
                                                      cropping_mode="strict")
            binarize = ref.blocks_utils.Binarize(0.5)
            pbuf, abuf = [], []
            for i in range(scores.shape[0]):
                sw = SlidingWindow(start=starts[i], duration=res, step=res)
                pbuf.append(SlidingWindowFeature(scores[i], sw))
                wav = (np.arange(80000, dtype=np.float64) + 8000.0 * i)[:, None]   # sample index ramp
                abuf.append(SlidingWindowFeature(wav, SlidingWindow(start=starts[i], duration=1 / 16000,
                                                                    step=1 / 16000)))
                agg, aud, mn = pred(pbuf), audio(abuf), mean(pbuf)
                ann = binarize(agg)
                turns = np.array([[seg.start, seg.end, spk] for seg, spk, _ in ann.itertracks(yield_label=True)],
                                 dtype=np.float64).reshape(-1, 3)
                tag = f"{tag0}_l{latency:g}_t{i}"
                out[tag + "_agg"] = agg.data
                out[tag + "_aggsw"] = np.array([agg.sliding_window.start, agg.sliding_window.step])
                out[tag + "_mean"] = mn.data
                out[tag + "_turns"] = turns
                out[tag + "_aud"] = np.array([aud.data.shape[0], aud.data[0, 0], aud.data[-1, 0],
                                              aud.sliding_window.start, aud.sliding_window.step])
                if len(pbuf) == pred.num_overlapping_windows:
                    pbuf, abuf = pbuf[1:], abuf[1:]
    np.savez_compressed(OUT / "tail.npz", **out)
    print("tail:", len(out), "arrays")


def pipelines():
    """The reference's OWN pipeline classes — blocks/diarization.py SpeakerDiarization and
    blocks/vad.py VoiceActivityDetection, loaded by path, unmodified — around the toy models of
    scenarios.py, driven like StreamingInference drives them (rolling 5 s windows, batches of
    consecutive chunks, optional timestamp shift).  Stored per case: what the reference's blocks
    handed to the clustering (segmentation, overlap-aware normalised embeddings) and what the
    pipeline returned per chunk (speech turns, a digest of the aggregated waveform)."""
    from oracle.pyannote_stub import load_reference_pipelines
    ref = load_reference_pipelines()
    stream = scenarios.pipeline_stream()
    S, H = 80000, 8000
    chunks = [SlidingWindowFeature(stream[i * H:i * H + S, None],
                                   SlidingWindow(start=i * 0.5, duration=1 / 16000, step=1 / 16000))
              for i in range((len(stream) - S) // H + 1)]
    out = {"num_chunks": np.array(len(chunks))}

    class Rec:
        def __init__(self, inner):
            self.inner, self.got = inner, []

        def __call__(self, *a):
            r = self.inner(*a)
            self.got.append(r.numpy().copy())
            return r

    def table(outputs):
        rows = []
        for i, (ann, _) in enumerate(outputs):
            for seg, _, label in ann.itertracks(yield_label=True):
                rows.append([i, seg.start, seg.end, float(label[len("speaker"):]) if label.startswith("speaker") else 0.0])
        return np.array(rows, dtype=np.float64).reshape(-1, 4)

    def digest(outputs):
        return np.array([[w.data.shape[0], w.sliding_window.start, w.sliding_window.step,
                          float(np.asarray(w.data, dtype=np.float64).sum()), w.data[0, 0], w.data[-1, 0]]
                         for _, w in outputs], dtype=np.float64)

    for name, (latency, bs, shift, tau, rho, delta) in scenarios.PIPE_CASES.items():
        seg_m = ref.models.SegmentationModel(lambda: scenarios.ToySegmentation())
        emb_m = ref.models.EmbeddingModel(lambda: scenarios.ToyEmbedding())
        cfg = ref.diarization.SpeakerDiarizationConfig(segmentation=seg_m, embedding=emb_m, latency=latency,
                                                       tau_active=tau, rho_update=rho, delta_new=delta,
                                                       device=torch.device("cpu"))
        pipe = ref.diarization.SpeakerDiarization(cfg)
        pipe.set_timestamp_shift(shift)
        pipe.segmentation, pipe.embedding = Rec(pipe.segmentation), Rec(pipe.embedding)
        outputs = []
        for i in range(0, len(chunks), bs):
            outputs += pipe(chunks[i:i + bs])
        out[f"{name}_seg"] = np.concatenate(pipe.segmentation.got)
        out[f"{name}_emb"] = np.concatenate([e if e.ndim == 3 else e[None] for e in pipe.embedding.got])
        out[f"{name}_turns"] = table(outputs)
        out[f"{name}_audio"] = digest(outputs)
        # VoiceActivityDetection with the same segmentation model (config 5's pipeline)
        vcfg = ref.vad.VoiceActivityDetectionConfig(segmentation=ref.models.SegmentationModel(lambda: scenarios.ToySegmentation()),
                                                    latency=latency, tau_active=tau, device=torch.device("cpu"))
        vad = ref.vad.VoiceActivityDetection(vcfg)
        vad.set_timestamp_shift(shift)
        vouts = []
        for i in range(0, len(chunks), bs):
            vouts += vad(chunks[i:i + bs])
        out[f"{name}_vad_turns"] = table(vouts)
        print(f"pipeline {name}: {len(chunks)} chunks, {len(out[name + '_turns'])} diarization turns, "
              f"{int(out[name + '_turns'][:, 3].max()) + 1} speakers, {len(out[name + '_vad_turns'])} speech regions")
    np.savez_compressed(OUT / "pipeline_toy.npz", **out)


def rttm_slice():
    """The first 40 lines of two meetings of the reference's expected_outputs/online/0.5s/AMI.rttm
    (paper-implementation output): an RTTM I/O fixture for features.load_rttm / Annotation.to_rttm."""
    src = Path("/root/reference/expected_outputs/online/0.5s/AMI.rttm")
    per, keep = {}, []
    for line in src.read_text().splitlines():
        uri = line.split()[1]
        if len(per) < 2 or uri in per:
            per[uri] = per.get(uri, 0) + 1
            if per[uri] <= 40:
                keep.append(line)
    (OUT / "ami_0.5s_slice.rttm").write_text("\n".join(keep) + "\n")
    print("rttm slice:", {u: min(n, 40) for u, n in per.items()})


if __name__ == "__main__":
    if "--rttm-only" in sys.argv:
        rttm_slice()
    elif "--tail-only" in sys.argv:
        tail(load_reference())
    elif "--pipelines-only" in sys.argv:
        pipelines()
    else:
        main()
        tail(load_reference())
        pipelines()
        rttm_slice()
