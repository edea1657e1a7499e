"""Golden output of the long-horizon scenario: 4 streams x 600 s (1 191 chunks each) through the REFERENCE'S OWN
pipeline — /root/reference/src/diart/blocks/diarization.py (SpeakerDiarization: its segmentation / embedding blocks,
OnlineSpeakerClustering, DelayedAggregation, Binarize), loaded by path with oracle/pyannote_stub.py — around the
restated networks of oracle/models_ref.py with the seeded weights of diart_amd/synth.py, latency 0.5 s and 5 s.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_long_horizon.py        (build container, ~6 min on 8 cores)

Writes tests/golden/long_horizon.npz: per (stream, latency) the speech turns of every chunk; per stream the
local -> global assignment of every chunk, the centroids at scenarios.LONG_CENTER_STEPS and the active-centre sets
(clustering does not depend on the latency).  The running-sum centroids of blocks/clustering.py:197-208 are what a
small numeric difference could drive apart over 1 191 steps: tests/test_gpu_long_horizon.py holds the GPU path to
this file."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE))

from oracle.models_ref import PyanNetRef, XVectorSincNetRef  # noqa: E402
from oracle.pyannote_stub import SlidingWindow, SlidingWindowFeature, load_reference_pipelines  # noqa: E402
from diart_amd.synth import synth_embedding_state, synth_segmentation_state  # noqa: E402
import scenarios  # noqa: E402


class Replay:
    """First pass: run the block and keep its outputs; later passes: hand the kept outputs back (the networks see
    the same chunks whatever the latency)."""

    def __init__(self, inner):
        self.inner, self.kept, self.pos, self.replay = inner, [], 0, False

    def __call__(self, *a):
        if self.replay and self.pos < len(self.kept):
            r = self.kept[self.pos]
            self.pos += 1
            return r
        r = self.inner(*a)
        self.kept.append(r)
        self.pos += 1
        return r

    def rewind(self):
        self.pos, self.replay = 0, True


def main():
    torch.set_num_threads(8)
    ref = load_reference_pipelines()
    seg_net, emb_net = PyanNetRef().eval(), XVectorSincNetRef().eval()
    seg_net.load_state_dict(synth_segmentation_state())
    emb_net.load_state_dict(synth_embedding_state())
    S, H, bs = 80000, 8000, 16
    out = {}
    for si in range(len(scenarios.LONG_STREAMS)):
        t0 = time.time()
        audio = scenarios.long_horizon_audio(si)
        n = (len(audio) - S) // H + 1
        chunks = [SlidingWindowFeature(audio[i * H:i * H + S, None], SlidingWindow(start=i * 0.5, duration=1 / 16000, step=1 / 16000))
                  for i in range(n)]
        replays = None
        for latency in scenarios.LONG_LATENCIES:
            cfg = ref.diarization.SpeakerDiarizationConfig(
                segmentation=ref.models.SegmentationModel(lambda: seg_net), embedding=ref.models.EmbeddingModel(lambda: emb_net),
                latency=latency, device=torch.device("cpu"))
            pipe = ref.diarization.SpeakerDiarization(cfg)
            if replays is None:
                replays = (Replay(pipe.segmentation), Replay(pipe.embedding))
            else:
                for r in replays:
                    r.rewind()
            pipe.segmentation, pipe.embedding = replays
            assign, centers, active = [], {}, {}
            clu = pipe.clustering
            ident = clu.identify

            def identify(segmentation, embeddings, ident=ident, clu=clu, assign=assign, centers=centers, active=active):
                m = ident(segmentation, embeddings)
                a = -np.ones(segmentation.data.shape[1], dtype=np.int8)
                for s_, t_ in zip(*m.valid_assignments()):
                    a[s_] = t_
                step = len(assign)
                assign.append(a)
                if step in scenarios.LONG_CENTER_STEPS:
                    centers[step] = clu.centers.astype(np.float32).copy()
                    active[step] = np.array([c in clu.active_centers for c in range(clu.max_speakers)])
                return m

            clu.identify = identify
            rows = []
            for i in range(0, n, bs):
                for j, (ann, _) in enumerate(pipe(chunks[i:i + bs])):
                    for seg, _, label in ann.itertracks(yield_label=True):
                        rows.append([i + j, seg.start, seg.end, float(label[len("speaker"):])])
            # What a FILE adds (Benchmark -> FileAudioSource with config.get_file_padding, blocks/diarization.py:
            # right padding = latency - step seconds of zeros, utils.get_padding_right): the windows that flush
            # the last `latency - step` seconds.  Rows with chunk index >= num_chunks; the streaming test ignores
            # them, the Benchmark test needs them.
            extra = int(round((latency - 0.5) / 0.5))
            if extra > 0:
                padded = np.concatenate([audio, np.zeros(extra * H, dtype=np.float32)])
                tail_chunks = [SlidingWindowFeature(padded[i * H:i * H + S, None],
                                                    SlidingWindow(start=i * 0.5, duration=1 / 16000, step=1 / 16000))
                               for i in range(n, n + extra)]
                for j, (ann, _) in enumerate(pipe(tail_chunks)):
                    for seg, _, label in ann.itertracks(yield_label=True):
                        rows.append([n + j, seg.start, seg.end, float(label[len("speaker"):])])
            out[f"turns_{si}_{latency}"] = np.array(rows, dtype=np.float64).reshape(-1, 4)
            if latency == scenarios.LONG_LATENCIES[0]:
                out[f"assign_{si}"] = np.stack(assign)
                out[f"centers_{si}"] = np.stack([centers[s] for s in scenarios.LONG_CENTER_STEPS])
                out[f"active_{si}"] = np.stack([active[s] for s in scenarios.LONG_CENTER_STEPS])
            else:
                assert np.array_equal(out[f"assign_{si}"], np.stack(assign[:n])), "clustering depended on the latency"
            print(f"stream {si} latency {latency}: {n} chunks, {len(rows)} turns, "
                  f"{int(out[f'active_{si}'][-1].sum())} global speakers, {time.time() - t0:.0f} s", flush=True)
    out["num_chunks"] = np.array(n)
    np.savez_compressed(HERE / "long_horizon.npz", **out)
    print("wrote", HERE / "long_horizon.npz", (HERE / "long_horizon.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
