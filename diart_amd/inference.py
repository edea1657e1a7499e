"""File-parallel evaluation of a pipeline — BASELINE.json configs 1 and 4.

The drivers on either side of the hot path, restated without ``rx`` / ``torchaudio``:

* ``file_blocks``       ``FileAudioSource.read``       reference ``sources.py:85-135``
* ``rolling_windows``   ``rearrange_audio_stream``     reference ``operators.py:44-100``
* ``StreamingInference`` (file sources, batched)       reference ``inference.py:22-253``
* ``PredictionAccumulator``                            reference ``sinks.py:59-88``
* ``Benchmark``                                        reference ``inference.py:255-432``
* ``DistributedBenchmark`` — what ``Parallelize`` (``inference.py:435-559``) does with a pool of
  spawned workers that each copy the models, done the MI355X way: one process per GPU
  (``torchrun``), whole files assigned to ranks by longest-processing-time (no file is split,
  no data-path collective), the weights broadcast once (``distributed.broadcast_state``) and the
  per-file error components gathered at the end.

Any ``blocks.Pipeline`` subclass drops in (``SpeakerDiarization``, ``VoiceActivityDetection``);
nothing here touches the GPU itself.
"""
from __future__ import annotations

import wave
from pathlib import Path
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import distributed as D
from .features import Annotation, SlidingWindow, SlidingWindowFeature, load_rttm

FilePath = Union[str, Path]


# ------------------------------------------------------------------------------------- audio
def read_wav_into(path: FilePath, alloc, padding_of, block_duration: float = 0.5):
    """``padded_file(read_wav(path))`` in ONE pass: decode the PCM samples straight into
    ``alloc(n_total)`` (e.g. a pinned upload buffer) at their padded position.  ``padding_of(duration
    seconds) -> (left, right)`` seconds.  -> (array, sample rate, (left, right)); same samples as the
    two-step form (the 2^-15 / 2^-31 / 2^-7 scalings are exact in float32)."""
    with wave.open(str(path), "rb") as f:
        sr, nch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    padding = padding_of(n / sr)
    left, right = (int(np.rint(p * sr)) for p in padding)
    size = int(np.rint(block_duration * sr))
    total = left + n + right
    out = alloc(-(-total // size) * size)
    out[:left] = 0.0
    out[left + n:] = 0.0
    dst = out[left:left + n]
    if nch == 1 and width == 2:
        np.multiply(np.frombuffer(raw, dtype="<i2"), np.float32(1.0 / 32768.0), out=dst, dtype=np.float32, casting="unsafe")
    else:
        x, _ = read_wav(path)
        dst[:] = x
    return out, sr, padding


def read_wav(path: FilePath) -> Tuple[np.ndarray, int]:
    """PCM WAV (8/16/32-bit integer) -> (mono float32 in [-1, 1], sample rate); channels are
    averaged like ``AudioLoader(mono=True)`` (reference ``audio.py:36-40``)."""
    with wave.open(str(path), "rb") as f:
        sr, nch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    x = x.reshape(-1, nch)
    return (x.mean(axis=1) if nch > 1 else x[:, 0]).astype(np.float32), sr


def write_wav(path: FilePath, samples: np.ndarray, sample_rate: int = 16000) -> None:
    """Mono float waveform -> 16-bit PCM WAV."""
    pcm = np.clip(np.rint(np.asarray(samples, dtype=np.float64) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sample_rate)
        f.writeframes(pcm.tobytes())


def wav_duration(path: FilePath) -> float:
    with wave.open(str(path), "rb") as f:
        return f.getnframes() / float(f.getframerate())


def file_blocks(waveform: np.ndarray, sample_rate: int, padding: Tuple[float, float] = (0, 0),
                block_duration: float = 0.5) -> Iterator[np.ndarray]:
    """Blocks ``(1, block_size)`` of a mono file as ``FileAudioSource.read`` emits them: zero
    padding left / right in seconds, a trailing incomplete block zero-filled (and up-cast to
    float64 by the concatenation with ``np.zeros``, as in the reference ``sources.py:117-121``)."""
    wav = np.asarray(waveform, dtype=np.float32).reshape(-1)
    left, right = (int(np.rint(p * sample_rate)) for p in padding)
    if left > 0:
        wav = np.concatenate([np.zeros(left, dtype=np.float32), wav])
    if right > 0:
        wav = np.concatenate([wav, np.zeros(right, dtype=np.float32)])
    size = int(np.rint(block_duration * sample_rate))
    full = wav.shape[0] // size
    for i in range(full):
        yield wav[None, i * size:(i + 1) * size]
    if wav.shape[0] % size != 0:
        last = wav[full * size:]
        yield np.concatenate([last[None, :], np.zeros((1, size - last.shape[0]))], axis=-1)


def padded_file(waveform: np.ndarray, sample_rate: int, padding: Tuple[float, float] = (0, 0),
                block_duration: float = 0.5) -> np.ndarray:
    """The concatenation of everything ``file_blocks`` emits, as one float32 array: zero padding
    left / right, the trailing incomplete block zero-filled.  (``file_blocks`` up-casts that last
    block to float64 like the reference; every consumer casts back with ``.float()``,
    ``features.py:124``, so float32 zeros are the same samples.)"""
    wav = np.asarray(waveform, dtype=np.float32).reshape(-1)
    left, right = (int(np.rint(p * sample_rate)) for p in padding)
    size = int(np.rint(block_duration * sample_rate))
    total = left + wav.shape[0] + right
    out = np.zeros(-(-total // size) * size, dtype=np.float32)
    out[left:left + wav.shape[0]] = wav
    return out


def rolling_windows(blocks: Iterable[np.ndarray], duration: float = 5.0, step: float = 0.5,
                    sample_rate: int = 16000) -> Iterator[SlidingWindowFeature]:
    """``rearrange_audio_stream``: buffer blocks until ``step`` seconds are available, append them
    to the current chunk, keep its last ``duration`` seconds (the start time advances by ``step``
    whenever samples are dropped) and emit every changed chunk that is exactly ``duration`` long,
    as a ``(samples, 1)`` feature whose frame grid starts at the chunk's start time."""
    chunk_samples, step_samples = int(round(sample_rate * duration)), int(round(sample_rate * step))
    chunk, buffer, start = None, None, 0
    for value in blocks:
        if value.ndim != 2 or value.shape[0] != 1:
            raise ValueError(f"Waveform must have shape (1, samples) but {value.shape} was found")
        buffer = value if buffer is None else np.concatenate([buffer, value], axis=1)
        if buffer.shape[1] < step_samples:
            continue
        new, buffer = (buffer, None) if buffer.shape[1] == step_samples else \
            (buffer[:, :step_samples], buffer[:, step_samples:])
        chunk = new if chunk is None else np.concatenate([chunk, new], axis=1)
        if chunk.shape[1] > chunk_samples:
            chunk = chunk[:, -chunk_samples:]
            start += step
        if chunk.shape[1] == chunk_samples:
            yield SlidingWindowFeature(chunk.T, SlidingWindow(start=start, duration=1.0 / sample_rate,
                                                              step=1.0 / sample_rate))


# -------------------------------------------------------------------------------- inference
class PredictionAccumulator:
    """Joins the per-chunk annotations of one stream; same-speaker turns closer than
    ``patch_collar`` are stitched (``Annotation.support``)."""

    def __init__(self, uri: Optional[str] = None, patch_collar: float = 0.05):
        self.uri, self.patch_collar = uri, patch_collar
        self._prediction: Optional[Annotation] = None

    def on_next(self, value):
        prediction = value[0] if isinstance(value, tuple) else value
        prediction.uri = self.uri
        if self._prediction is None:
            self._prediction = prediction
        else:
            self._prediction.update(prediction)

    def get_prediction(self) -> Optional[Annotation]:
        if self._prediction is not None:
            self._prediction = self._prediction.support(self.patch_collar)
        return self._prediction


class StreamingInference:
    """One file through one pipeline: blocks -> rolling windows -> batches of ``batch_size``
    consecutive windows -> ``pipeline(batch)`` -> accumulated ``Annotation``."""

    def __init__(self, pipeline, waveform: np.ndarray, sample_rate: int, uri: str = "stream",
                 padding: Tuple[float, float] = (0, 0), batch_size: int = 1, hooks: Sequence = ()):
        cfg = pipeline.config
        if sample_rate != cfg.sample_rate:
            raise ValueError(f"audio source has sample rate {sample_rate}, the pipeline's is "
                             f"{cfg.sample_rate}; resample the file first")
        self.pipeline, self.batch_size, self.hooks = pipeline, max(1, int(batch_size)), list(hooks)
        self.accumulator = PredictionAccumulator(uri)
        self._windows = rolling_windows(file_blocks(waveform, sample_rate, padding, cfg.step),
                                        cfg.duration, cfg.step, sample_rate)
        total = padding[0] + len(waveform) / sample_rate + padding[1]
        self.num_chunks = int(np.ceil((total - cfg.duration + cfg.step) / cfg.step))
        self.chunks_done = 0

    def __call__(self) -> Optional[Annotation]:
        batch: List[SlidingWindowFeature] = []

        def flush():
            for out in self.pipeline(batch):
                self.accumulator.on_next(out)
                for hook in self.hooks:
                    hook(out)
            self.chunks_done += len(batch)
            batch.clear()

        for window in self._windows:
            batch.append(window)
            if len(batch) == self.batch_size:
                flush()
        if batch:
            flush()
        return self.accumulator.get_prediction()


class Benchmark:
    """Run a pipeline class on every WAV of ``speech_path``; write ``<uri>.rttm`` files to
    ``output_path`` and / or score against ``reference_path/<uri>.rttm``.  Same constructor and call
    as the reference's; the report is the metric object (``abs(metric)`` = aggregate rate,
    ``metric.report()`` = the per-file table) instead of a pandas frame of pyannote.metrics."""

    def __init__(self, speech_path: FilePath, reference_path: Optional[FilePath] = None,
                 output_path: Optional[FilePath] = None, show_progress: bool = False,
                 show_report: bool = True, batch_size: int = 32, concurrent_files: Optional[int] = None):
        self.speech_path = Path(speech_path).expanduser()
        assert self.speech_path.is_dir(), "Speech path must be a directory"
        assert reference_path is not None or output_path is not None, \
            "Benchmark expected reference path, output path or both"
        self.reference_path = None
        if reference_path is not None:
            self.reference_path = Path(reference_path).expanduser()
            assert self.reference_path.is_dir(), "Reference path must be a directory"
        self.output_path = None
        if output_path is not None:
            self.output_path = Path(output_path).expanduser()
            self.output_path.mkdir(parents=True, exist_ok=True)
        self.show_progress, self.show_report, self.batch_size = show_progress, show_report, batch_size
        # files of this process that share one GPU batch (``FileBatch``); None = 16, 0 = the reference's
        # one-file-at-a-time loop.  Only the HIP x-vector diarization pipeline has the batched path; anything else
        # falls back to the loop.
        from .config import setting
        self.concurrent_files = int(setting("concurrent_files", concurrent_files, 16, int))
        self.last_path = None      # "file_batch" | "one_file_at_a_time": which path the last call took

    def get_file_paths(self) -> List[Path]:
        return sorted(p for p in self.speech_path.iterdir() if p.suffix.lower() == ".wav")

    def run_single(self, pipeline, filepath: Path) -> Annotation:
        """Does NOT reset the pipeline (like the reference's ``run_single``)."""
        waveform, sr = read_wav(filepath)
        padding = pipeline.config.get_padding(len(waveform) / sr)
        pipeline.set_timestamp_shift(-padding[0])
        pred = StreamingInference(pipeline, waveform, sr, filepath.stem, padding, self.batch_size)()
        if pred is None:
            pred = Annotation(filepath.stem)
        pred.uri = filepath.stem
        if self.output_path is not None:
            with open(self.output_path / f"{filepath.stem}.rttm", "w") as out_file:
                pred.write_rttm(out_file)
        if self.show_progress:
            print(f"[benchmark] {filepath.stem}: {len(pred)} turns", flush=True)
        return pred

    def file_batch(self, pipeline_class: type, config):
        """A ``FileBatch`` for (pipeline_class, config), or None when this combination has to go
        through the one-file-at-a-time loop (custom pipeline classes, non-HIP or ECAPA models, CPU)."""
        from .blocks.diarization import SpeakerDiarization
        from .models import HipEmbedding, HipSegmentation
        if self.concurrent_files <= 0 or pipeline_class is not SpeakerDiarization:
            return None
        if getattr(config.device, "type", "cpu") != "cuda":
            return None
        seg, emb = config.segmentation, config.embedding
        seg.load()
        emb.load()
        if type(seg.model) is not HipSegmentation or type(emb.model) is not HipEmbedding:
            return None
        from .pipeline import FileBatch
        # one engine per (models, hyper-parameters): its scratch arenas (~1 GB) and pinned slots are
        # allocated once, not per call
        key = (id(seg.model), id(emb.model), config.tau_active, config.rho_update, config.delta_new, config.gamma,
               config.beta, config.max_speakers, config.normalize_embedding_weights, config.duration, config.step,
               config.latency, config.sample_rate, str(config.device), self.batch_size, self.concurrent_files)
        cached = getattr(self, "_fb_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        fb = self._new_file_batch(FileBatch, seg, emb, config)
        self._fb_cache = (key, fb)
        return fb

    def _new_file_batch(self, FileBatch, seg, emb, config):
        return FileBatch(seg.model, emb.model, rows=max(64, self.batch_size), max_files=self.concurrent_files,
                         tau_active=config.tau_active, rho_update=config.rho_update, delta_new=config.delta_new,
                         gamma=config.gamma, beta=config.beta, max_speakers=config.max_speakers,
                         normalize_embedding_weights=config.normalize_embedding_weights,
                         duration=config.duration, step=config.step, latency=config.latency,
                         sample_rate=config.sample_rate, device=config.device)

    def run_batched(self, fb, config, paths: Sequence[Path]) -> List[Annotation]:
        """``paths`` through one ``FileBatch``: same padding, timestamp shift, RTTM files and
        progress lines as ``run_single``, files read lazily as slots free up."""
        def feed():      # runs on the FileBatch's loader thread: decode straight into pinned memory
            for fp in paths:
                padded, sr, padding = read_wav_into(fp, fb.host_buffer, config.get_padding, config.step)
                if sr != config.sample_rate:
                    raise ValueError(f"audio source has sample rate {sr}, the pipeline's is "
                                     f"{config.sample_rate}; resample the file first")
                yield fp.stem, padded, -padding[0]

        got = fb.run(feed())
        preds = []
        for fp in paths:
            pred = got[fp.stem]
            pred.uri = fp.stem
            if self.output_path is not None:
                with open(self.output_path / f"{fp.stem}.rttm", "w") as out_file:
                    pred.write_rttm(out_file)
            if self.show_progress:
                print(f"[benchmark] {fp.stem}: {len(pred)} turns", flush=True)
            preds.append(pred)
        return preds

    def evaluate(self, predictions: List[Annotation], metric):
        if self.reference_path is None:
            return predictions
        for hyp in predictions:
            ref = load_rttm(self.reference_path / f"{hyp.uri}.rttm").popitem()[1]
            metric(ref, hyp)
        if self.show_report:
            print(metric.report(), flush=True)
        return metric

    def __call__(self, pipeline_class: type, config, metric=None):
        pipeline = pipeline_class(config)
        fb = self.file_batch(pipeline_class, config)
        self.last_path = "file_batch" if fb is not None else "one_file_at_a_time"
        if fb is not None:
            predictions = self.run_batched(fb, config, self.get_file_paths())
        else:
            predictions = []
            for filepath in self.get_file_paths():
                pipeline.reset()
                predictions.append(self.run_single(pipeline, filepath))
        metric = pipeline.suggest_metric() if metric is None else metric
        return self.evaluate(predictions, metric)


class DistributedBenchmark:
    """``Benchmark`` over the ranks of a ``torchrun`` job (one process per GPU): rank r runs the
    files ``shard_files_lpt`` gives it, every rank writes its own RTTMs, and rank 0 receives the
    per-file error components of all ranks (``all_gather_object``: a few doubles per file — the
    only communication besides the one-time weight broadcast).  With WORLD_SIZE = 1 it is
    ``Benchmark``."""

    def __init__(self, benchmark: Benchmark):
        self.benchmark = benchmark

    def __call__(self, pipeline_class: type, config, metric=None):
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        b = self.benchmark
        files = b.get_file_paths()
        mine = D.shard_files_lpt([wav_duration(p) for p in files], world)[rank]
        pipeline = pipeline_class(config)
        metric = pipeline.suggest_metric() if metric is None else metric
        local: Dict[str, Optional[dict]] = {}
        fb = b.file_batch(pipeline_class, config)
        b.last_path = "file_batch" if fb is not None else "one_file_at_a_time"
        # longest first: the short files fill the slots the long ones leave at the end
        batched = {h.uri: h for h in b.run_batched(fb, config, [files[i] for i in mine])} if fb is not None else None
        for i in sorted(mine):
            if batched is not None:
                hyp = batched[files[i].stem]
            else:
                pipeline.reset()
                hyp = b.run_single(pipeline, files[i])
            comp = None
            if b.reference_path is not None:
                ref = load_rttm(b.reference_path / f"{hyp.uri}.rttm").popitem()[1]
                comp = metric.components(ref, hyp)
            local[hyp.uri] = comp
        gathered = [local]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, local)
        merged: Dict[str, Optional[dict]] = {}
        for part in gathered:
            merged.update(part)
        if b.reference_path is None:
            return sorted(merged)
        metric.reset()
        for uri in sorted(merged):
            for c, v in merged[uri].items():
                metric.accumulated[c] += v
            metric.results.append((uri, merged[uri]))
        if b.show_report and rank == 0:
            print(metric.report(), flush=True)
        return metric
