"""Host logic of the file-parallel evaluation (BASELINE.json configs 1 / 4): WAV I/O, the block
source and rolling windows (reference sources.py:85-135, operators.py:44-100), the batched
streaming loop, ``Benchmark`` and its rank-sharded form over a world_size-2 gloo group.  The
pipeline here is a tiny CPU energy detector honouring the ``blocks.Pipeline`` contract; the real
pipelines go through the same drivers in tests/test_gpu_der.py."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from diart_amd import inference as I
from diart_amd.blocks import base
from diart_amd.features import Annotation, Segment, SlidingWindowFeature

ROOT = Path(__file__).resolve().parent.parent

PIPELINE_SRC = r'''
import numpy as np
from diart_amd.blocks import base
from diart_amd.features import Annotation, Segment
from diart_amd.metrics import DetectionErrorRate


class EnergyConfig(base.PipelineConfig):
    def __init__(self, step=0.5, latency=None, duration=5.0, sample_rate=16000):
        self._d, self._s, self._sr = duration, step, sample_rate
        self._l = step if latency is None else latency

    duration = property(lambda self: self._d)
    step = property(lambda self: self._s)
    latency = property(lambda self: self._l)
    sample_rate = property(lambda self: self._sr)


class EnergyVAD(base.Pipeline):
    """Speech = the newest `step` seconds of the window have RMS > 0.05 (one turn per chunk)."""

    def __init__(self, config=None):
        self._config = EnergyConfig() if config is None else config
        self.shift, self.calls, self.resets = 0.0, [], 0

    get_config_class = staticmethod(lambda: EnergyConfig)
    suggest_metric = staticmethod(lambda: DetectionErrorRate())
    hyper_parameters = staticmethod(lambda: [])
    config = property(lambda self: self._config)

    def reset(self):
        self.resets += 1
        self.shift = 0.0

    def set_timestamp_shift(self, shift):
        self.shift = shift

    def __call__(self, waveforms):
        self.calls.append(len(waveforms))
        out = []
        n = int(round(self.config.step * self.config.sample_rate))
        for w in waveforms:
            assert w.data.shape == (int(round(self.config.duration * self.config.sample_rate)), 1)
            end = w.extent.end
            ann = Annotation("x", "speech")
            if float(np.sqrt(np.mean(np.square(w.data[-n:])))) > 0.05:
                ann[Segment(end - self.config.step + self.shift, end + self.shift), 0] = "speech"
            out.append((ann, w))
        return out
'''
exec(PIPELINE_SRC)


def _tone(seconds, on, sr=16000, seed=0):
    """Noise bursts on the given (start, end) intervals, silence elsewhere."""
    rng = np.random.default_rng(seed)
    x = np.zeros(int(seconds * sr), dtype=np.float32)
    for a, b in on:
        x[int(a * sr):int(b * sr)] = 0.3 * rng.standard_normal(int(b * sr) - int(a * sr)).astype(np.float32)
    return np.clip(x, -1, 1)


def test_wav_roundtrip_and_blocks(tmp_path):
    x = _tone(3.3, [(0.5, 1.5)])
    I.write_wav(tmp_path / "a.wav", x)
    y, sr = I.read_wav(tmp_path / "a.wav")
    assert sr == 16000 and y.shape == x.shape and np.abs(y - x).max() <= 1 / 32768 + 1e-7
    assert abs(I.wav_duration(tmp_path / "a.wav") - 3.3) < 1e-9
    blocks = list(I.file_blocks(y, sr, padding=(0.25, 0.1), block_duration=0.5))
    # 0.25 + 3.3 + 0.1 = 3.65 s -> 7 full blocks + a zero-filled one, the last up-cast to float64
    assert len(blocks) == 8 and all(b.shape == (1, 8000) for b in blocks)
    assert blocks[0].dtype == np.float32 and blocks[-1].dtype == np.float64
    assert np.all(blocks[0][0, :4000] == 0) and np.array_equal(blocks[0][0, 4000:], y[:4000])
    assert np.all(blocks[-1][0, int(0.15 * 16000):] == 0)


def test_rolling_windows_follow_rearrange_audio_stream():
    sr, x = 16000, np.arange(30 * 16000, dtype=np.float32)
    wins = list(I.rolling_windows(I.file_blocks(x, sr, (0, 0), 0.5), 5.0, 0.5, sr))
    assert len(wins) == 51                                   # ceil((30 - 5 + 0.5) / 0.5), inference.py:81-83
    for i, w in enumerate(wins):
        assert isinstance(w, SlidingWindowFeature) and w.data.shape == (80000, 1)
        assert w.data[0, 0] == i * 8000 and w.data[-1, 0] == i * 8000 + 79999
        assert abs(w.sliding_window.start - 0.5 * i) < 1e-9 and abs(w.sliding_window.step - 1 / sr) < 1e-15
    # blocks that do not divide the step are buffered; a short stream emits nothing
    odd = [x[None, i:i + 3000] for i in range(0, 120000, 3000)]
    w2 = list(I.rolling_windows(odd, 5.0, 0.5, sr))
    assert len(w2) == (120000 - 80000) // 8000 + 1 and np.array_equal(w2[3].data, wins[3].data)
    assert list(I.rolling_windows([x[None, :70000]], 5.0, 0.5, sr)) == []
    with pytest.raises(ValueError):
        list(I.rolling_windows([x[:100]], 5.0, 0.5, sr))


def test_streaming_inference_batches_and_padding():
    x = _tone(12.0, [(2.0, 4.0), (7.0, 7.5), (8.0, 8.5), (8.52, 9.0)])
    pipe = EnergyVAD(EnergyConfig(latency=2.0))
    padding = pipe.config.get_padding(12.0)
    assert padding == (0, 1.5)                               # right = latency - step (utils.py:87-88)
    pipe.set_timestamp_shift(-padding[0])
    inf = I.StreamingInference(pipe, x, 16000, "f1", padding, batch_size=4)
    assert inf.num_chunks == int(np.ceil((13.5 - 5 + 0.5) / 0.5))
    pred = inf()
    assert inf.chunks_done == inf.num_chunks and pipe.calls == [4, 4, 4, 4, 2]
    turns = [(round(s.start, 3), round(s.end, 3)) for s, _, _ in pred.itertracks(yield_label=True)]
    # windows start emitting at t = 5 s: the burst at 2-4 s is never "the newest half second";
    # 7.0-7.5 and 8.0-... stay apart (the gap is a whole step), the per-chunk turns 8.0-8.5 and
    # 8.5-9.0 are stitched by PredictionAccumulator's support(0.05)
    assert pred.uri == "f1" and turns == [(7.0, 7.5), (8.0, 9.0)]
    short = EnergyVAD(EnergyConfig())
    assert short.config.get_padding(3.0) == (2.0, 0.0)       # left-pad up to one chunk (utils.py:69-72)
    with pytest.raises(ValueError):
        I.StreamingInference(short, x, 8000)


def _make_corpus(tmp_path, n=5):
    speech, refs = tmp_path / "wav", tmp_path / "ref"
    speech.mkdir()
    refs.mkdir()
    for i in range(n):
        dur = 8.0 + 2.0 * i
        on = [(5.5 + 0.5 * i, 7.0 + 0.5 * i)]
        I.write_wav(speech / f"file{i}.wav", _tone(dur, on, seed=i))
        ref = Annotation(f"file{i}", "speech")
        ref[Segment(*on[0]), 0] = "speech"
        with open(refs / f"file{i}.rttm", "w") as f:
            ref.write_rttm(f)
    (speech / "notes.txt").write_text("not audio")
    return speech, refs


def test_benchmark_writes_rttm_and_scores(tmp_path):
    speech, refs = _make_corpus(tmp_path)
    out = tmp_path / "out"
    bench = I.Benchmark(speech, refs, out, show_report=False, batch_size=32)
    assert [p.name for p in bench.get_file_paths()] == [f"file{i}.wav" for i in range(5)]
    metric = bench(EnergyVAD, EnergyConfig())
    assert len(metric.results) == 5 and abs(metric) < 1e-9   # the detector is exact on these files
    assert sorted(p.name for p in out.iterdir()) == [f"file{i}.rttm" for i in range(5)]
    line = (out / "file2.rttm").read_text().splitlines()[0]
    assert line == "SPEAKER file2 1 6.500 1.500 <NA> <NA> speech <NA> <NA>"
    preds = I.Benchmark(speech, None, out, show_report=False)(EnergyVAD, EnergyConfig())
    assert [p.uri for p in preds] == [f"file{i}" for i in range(5)]
    with pytest.raises(AssertionError):
        I.Benchmark(speech)
    with pytest.raises(AssertionError):
        I.Benchmark(tmp_path / "missing", refs)


WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1])
from diart_amd import distributed as D, inference as I
exec(open(sys.argv[5]).read())
rank, world, local = D.init_from_env("gloo")
bench = I.Benchmark(sys.argv[2], sys.argv[3], sys.argv[4], show_report=False, batch_size=8)
metric = I.DistributedBenchmark(bench)(EnergyVAD, EnergyConfig())
assert [u for u, _ in metric.results] == [f"file{i}" for i in range(5)], metric.results
assert abs(metric) < 1e-9 and metric.accumulated["total"] == 7.5
import torch.distributed as dist
dist.barrier()
print("rank", rank, "ok")
'''


def test_distributed_benchmark_world_size_2_gloo(tmp_path):
    speech, refs = _make_corpus(tmp_path)
    out = tmp_path / "out"
    (tmp_path / "pipeline.py").write_text(PIPELINE_SRC)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_host_logic import _two_ranks
    for rank, (p, o) in enumerate(_two_ranks(script, [ROOT, speech, refs, out, tmp_path / "pipeline.py"])):
        assert p.returncode == 0, o
        assert f"rank {rank} ok" in o
    # every file was written exactly once, by the rank that owned it
    assert sorted(p.name for p in out.iterdir()) == [f"file{i}.rttm" for i in range(5)]


def test_padded_file_is_the_concatenation_of_file_blocks():
    """FileBatch keeps every file of a rank resident as ONE array and reads consecutive windows in
    place: that array must be exactly what FileAudioSource emits block by block (sources.py:85-135)
    and its windows exactly rearrange_audio_stream's (operators.py:44-100), incl. left padding of a
    short file and the zero-filled last block."""
    from diart_amd.inference import file_blocks, padded_file, rolling_windows
    rng = np.random.default_rng(3)
    for n, padding in ((80000, (0, 0)), (123457, (0, 1.5)), (30000, (3.125, 0)), (8000, (4.5, 0.0)), (95999, (0, 0))):
        wav = rng.standard_normal(n).astype(np.float32)
        whole = padded_file(wav, 16000, padding, 0.5)
        blocks = np.concatenate([b.astype(np.float32) for b in file_blocks(wav, 16000, padding, 0.5)], axis=1)[0]
        assert whole.dtype == np.float32 and np.array_equal(whole, blocks)
        wins = list(rolling_windows(file_blocks(wav, 16000, padding, 0.5), 5.0, 0.5, 16000))
        assert len(wins) == max(0, (len(whole) - 80000) // 8000 + 1)
        start = 0.0
        for i, w in enumerate(wins):
            assert np.array_equal(w.data[:, 0].astype(np.float32), whole[i * 8000:i * 8000 + 80000])
            assert w.sliding_window.start == start
            start += 0.5


def test_read_wav_into_equals_read_then_pad(tmp_path):
    from diart_amd.inference import padded_file, read_wav, read_wav_into, write_wav
    rng = np.random.default_rng(4)
    for n in (30000, 81234, 160000):
        p = tmp_path / f"f{n}.wav"
        write_wav(p, rng.uniform(-1, 1, n), 16000)
        pad_of = lambda d: (max(0.0, 5.0 - d - 1.5), 1.5)   # noqa: E731
        got, sr, padding = read_wav_into(p, lambda m: np.full(m, 7.0, dtype=np.float32), pad_of, 0.5)
        wav, sr2 = read_wav(p)
        assert sr == sr2 == 16000 and padding == pad_of(n / 16000)
        assert np.array_equal(got, padded_file(wav, sr, padding, 0.5))
