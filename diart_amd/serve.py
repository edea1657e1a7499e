"""Many live streams on one GPU: a cross-stream batcher on top of ``StreamBatch``
(SURVEY.md §8f rank 4).

The reference serves ONE stream per process (``diart.serve`` / ``diart.stream``:
``console/serve.py:105-127``, ``sources.py:204-271`` feed a single ``StreamingInference``); its
per-stream chain is ``source blocks -> rearrange_audio_stream -> pipeline([chunk])``
(``inference.py:101-147``).  ``StreamServer`` keeps that chain per stream — arbitrary-length audio
blocks in, a new 5 s window every 500 ms of audio (``operators.py:44-100``) — but runs the windows
that are ready *across streams* as one GPU batch: every ``step()`` takes at most one pending
window from each open stream, stacks them, runs segmentation / embedding once and steps each
stream's own clustering and aggregation state.  Streams join and leave at any time and advance at
their own pace; per stream the output is what a dedicated ``SpeakerDiarization`` pipeline with the
same configuration produces.

Transport (websocket, microphone) is out of scope: ``push`` is the seam a network front-end calls
(thread-safe), ``step`` / ``serve_forever`` is the worker loop."""
from __future__ import annotations

import threading
import time
from typing import Callable, Dict, Hashable, List, Optional

import numpy as np
import torch

from .blocks.aggregation import BatchedOutputTail
from .features import Annotation
from .pipeline import StreamBatch


class _Stream:
    __slots__ = ("slot", "buffer", "chunk", "start", "emitted", "pending", "prediction")

    def __init__(self, slot: int):
        self.slot = slot
        self.buffer = np.zeros(0, dtype=np.float32)   # samples not yet part of a window step
        self.chunk: Optional[np.ndarray] = None       # the current (<= duration) window
        self.start = 0.0                              # start time of `chunk`
        self.emitted = 0
        self.pending: List = []                       # (window copy, start time), oldest first
        self.prediction: Optional[Annotation] = None


class StreamServer:
    def __init__(self, segmentation, embedding, max_streams: int = 64, *, duration: float = 5.0,
                 step: float = 0.5, latency: Optional[float] = None, sample_rate: int = 16000,
                 tau_active: float = 0.6, rho_update: float = 0.3, delta_new: float = 1.0,
                 gamma: float = 3, beta: float = 10, max_speakers: int = 20,
                 device: Optional[torch.device] = None, patch_collar: float = 0.05,
                 engine: Optional[Callable] = None):
        self.duration, self.step_seconds, self.sample_rate = float(duration), float(step), int(sample_rate)
        self.latency = self.step_seconds if latency is None else float(latency)
        self.chunk_samples = int(round(sample_rate * duration))
        self.step_samples = int(round(sample_rate * step))
        self.max_streams, self.patch_collar = int(max_streams), patch_collar
        self._lock = threading.Lock()          # stream table, buffers, slot lists
        self._step_lock = threading.Lock()     # serialises step(): one engine
        self._streams: Dict[Hashable, _Stream] = {}
        self._free = list(range(self.max_streams - 1, -1, -1))
        self._inflight: set = set()            # slots whose window the worker is processing now
        self._deferred: List[int] = []         # slots closed while in flight: freed after the step
        self._stop = False
        # `engine(windows (k, S) float32, starts (k,), slots [k]) -> list of k (turns (m, 3) array)`;
        # the default engine is a StreamBatch with the C++ output tail
        if engine is None:
            self.batch = StreamBatch(segmentation, embedding, self.max_streams, tau_active, rho_update,
                                     delta_new, gamma, beta, max_speakers, device=device, tail=True,
                                     duration=duration, step=step, latency=self.latency)
            self._pinned = torch.empty((self.max_streams, self.chunk_samples), dtype=torch.float32).pin_memory()
            self._dev = torch.empty((self.max_streams, self.chunk_samples), dtype=torch.float32,
                                    device=self.batch.device)
            self._engine, self._reset_slot = self._gpu_engine, self.batch.reset
        else:
            self.batch = None
            self._engine, self._reset_slot = engine, getattr(engine, "reset", lambda slot: None)

    # ------------------------------------------------------------------ stream life cycle
    def open(self, stream_id: Hashable) -> None:
        with self._lock:
            if stream_id in self._streams:
                raise ValueError(f"stream {stream_id!r} is already open")
            if not self._free:
                raise RuntimeError(f"all {self.max_streams} stream slots are in use")
            # a free slot is never in flight (close() defers the release of a slot the worker is
            # still stepping), so resetting its clustering / aggregation state here is safe
            slot = self._free.pop()
            self._reset_slot(slot)
            self._streams[stream_id] = _Stream(slot)

    def close(self, stream_id: Hashable) -> Annotation:
        """Drop the stream (windows still pending are discarded: call ``drain`` first to flush them)
        and return everything it said so far, stitched like ``PredictionAccumulator``
        (``sinks.py:59-88``)."""
        with self._lock:
            st = self._streams.pop(stream_id)
            if st.slot in self._inflight:
                # the worker is inside step() with a window of this stream: its clustering / tail
                # state is being read on host threads right now.  The slot is handed back (and may
                # then be reset by the next open()) only when that step has finished; the step's
                # result for this stream is discarded.
                self._deferred.append(st.slot)
            else:
                self._free.append(st.slot)
        pred = st.prediction if st.prediction is not None else Annotation(str(stream_id), "speech")
        pred.uri = str(stream_id)
        return pred.support(self.patch_collar)

    @property
    def open_streams(self) -> List[Hashable]:
        with self._lock:
            return list(self._streams)

    # ------------------------------------------------------------------ audio in
    def push(self, stream_id: Hashable, samples) -> int:
        """Append mono float samples (any length) to a stream; returns the number of windows now
        pending for it.  The windowing is ``rearrange_audio_stream`` (``operators.py:44-100``)."""
        x = np.asarray(samples, dtype=np.float32).reshape(-1)
        with self._lock:
            st = self._streams[stream_id]
            st.buffer = np.concatenate([st.buffer, x]) if st.buffer.size else x.copy()
            while st.buffer.size >= self.step_samples:
                new, st.buffer = st.buffer[:self.step_samples], st.buffer[self.step_samples:]
                st.chunk = new if st.chunk is None else np.concatenate([st.chunk, new])
                if st.chunk.size > self.chunk_samples:
                    st.chunk = st.chunk[-self.chunk_samples:]
                    st.start += self.step_seconds
                if st.chunk.size == self.chunk_samples:
                    st.pending.append((st.chunk.copy(), st.start))
            return len(st.pending)

    # ------------------------------------------------------------------ the worker
    def step(self) -> Dict[Hashable, Annotation]:
        """Process at most one pending window of every open stream, as ONE batch.  Returns the
        speech turns each of those streams gained (the per-chunk ``Annotation`` of the reference's
        pipeline); the running total is kept per stream until ``close``."""
        with self._step_lock:                       # one engine, one step at a time
            with self._lock:
                ready = [(sid, st) for sid, st in self._streams.items() if st.pending]
                if not ready:
                    return {}
                work = [(sid, st, *st.pending.pop(0)) for sid, st in ready]
                self._inflight = {st.slot for _, st, _, _ in work}
            try:
                windows = np.stack([w for _, _, w, _ in work])
                starts = np.array([t for _, _, _, t in work], dtype=np.float64)
                slots = [st.slot for _, st, _, _ in work]
                turns = self._engine(windows, starts, slots)
            finally:
                with self._lock:
                    self._inflight = set()
                    self._free.extend(self._deferred)     # slots closed while they were in flight
                    self._deferred = []
            out = {}
            with self._lock:
                for (sid, st, _, _), tr in zip(work, turns):
                    if self._streams.get(sid) is not st:      # closed (or re-opened) mid-step: drop
                        continue
                    ann = BatchedOutputTail.annotation(np.asarray(tr, dtype=np.float64).reshape(-1, 3),
                                                       len(tr), uri=str(sid))
                    st.emitted += 1
                    if st.prediction is None:
                        st.prediction = ann
                    else:
                        st.prediction.update(ann)
                    out[sid] = ann
            return out

    def drain(self) -> int:
        """``step`` until no window is pending; returns the number of steps."""
        n = 0
        while self.step():
            n += 1
        return n

    def serve_forever(self, idle_sleep: float = 0.002) -> None:
        while not self._stop:
            if not self.step():
                time.sleep(idle_sleep)

    def shutdown(self) -> None:
        self._stop = True

    # ------------------------------------------------------------------ GPU engine
    def _gpu_engine(self, windows: np.ndarray, starts: np.ndarray, slots: List[int]):
        k = windows.shape[0]
        self._pinned[:k].copy_(torch.from_numpy(windows))
        self._dev[:k].copy_(self._pinned[:k], non_blocking=True)
        ticket = self.batch.launch(self._dev[:k], starts, slots=slots)
        self.batch.finish(ticket, want_scores=False)
        _, _, _, _, turns, nturns = ticket["tail"]
        return [turns[i, :int(nturns[i])].copy() for i in range(k)]
