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

Audio reaches the GPU through per-stream device rings (``AudioRing.push_rows`` / ``gather``): a step
uploads only the NEW 500 ms block of each stream that has one (32 KB instead of the 320 KB window the
reference moves per chunk, ``blocks/segmentation.py:47``) and assembles the batch of windows on the
device.  ``device_rings=False`` (and any custom ``engine``) keeps the windows on the host instead.

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
from .pipeline import AudioRing, StreamBatch


class _Stream:
    __slots__ = ("slot", "buffer", "blocks", "consumed", "chunk", "start", "emitted", "prediction")

    def __init__(self, slot: int):
        self.slot = slot
        self.buffer = np.zeros(0, dtype=np.float32)   # samples not yet a whole step block
        self.blocks: List[np.ndarray] = []            # whole step blocks the worker has not taken yet
        self.consumed = 0                             # step blocks taken by the worker so far
        self.chunk: Optional[np.ndarray] = None       # host-window mode: the current (<= duration) window
        self.start = 0.0                              # host-window mode: start time of `chunk`
        self.emitted = 0
        self.prediction: Optional[Annotation] = None


class StreamServer:
    def __init__(self, segmentation, embedding, max_streams: int = 64, *, duration: float = 5.0,
                 step: float = 0.5, latency: Optional[float] = None, sample_rate: int = 16000,
                 tau_active: float = 0.6, rho_update: float = 0.3, delta_new: float = 1.0,
                 gamma: float = 3, beta: float = 10, max_speakers: int = 20,
                 device: Optional[torch.device] = None, patch_collar: float = 0.05,
                 engine: Optional[Callable] = None, device_rings: bool = True):
        self.duration, self.step_seconds, self.sample_rate = float(duration), float(step), int(sample_rate)
        self.latency = self.step_seconds if latency is None else float(latency)
        self.chunk_samples = int(round(sample_rate * duration))
        self.step_samples = int(round(sample_rate * step))
        self.max_streams, self.patch_collar = int(max_streams), patch_collar
        self.blocks_per_window = -(-self.chunk_samples // self.step_samples)
        self._lock = threading.Lock()          # stream table, buffers, slot lists
        self._step_lock = threading.Lock()     # serialises step(): one engine
        self._streams: Dict[Hashable, _Stream] = {}
        self._free = list(range(self.max_streams - 1, -1, -1))
        self._inflight: set = set()            # slots whose window the worker is processing now
        self._deferred: List[int] = []         # slots closed while in flight: freed after the step
        self._stop = False
        import collections
        self.step_errors = collections.deque(maxlen=64)   # (time, repr) of failed steps, newest last
        # `engine(windows (k, S) float32, starts (k,), slots [k]) -> list of k (turns (m, 3) array)`;
        # the default engine is a StreamBatch with the C++ output tail
        if engine is None:
            self.batch = StreamBatch(segmentation, embedding, self.max_streams, tau_active, rho_update,
                                     delta_new, gamma, beta, max_speakers, device=device, tail=True,
                                     duration=duration, step=step, latency=self.latency)
            self._dev = torch.empty((self.max_streams, self.chunk_samples), dtype=torch.float32,
                                    device=self.batch.device)
            self._engine, self._reset_slot = self._gpu_engine, self.batch.reset
            # per-stream device rings need whole blocks per window (and 16-byte aligned rows)
            self.rings: Optional[AudioRing] = None
            if device_rings and self.chunk_samples % self.step_samples == 0 and self.step_samples % 4 == 0:
                self.rings = AudioRing(self.max_streams, self.chunk_samples, self.step_samples, slack_blocks=0,
                                       device=self.batch.device)
                # a stream takes at most blocks_per_window blocks in one step (its warm-up); round r of
                # a step stages its blocks in plane r, so no plane is rewritten while the GPU reads it
                self._stage = torch.empty((self.blocks_per_window, self.max_streams, self.step_samples),
                                          dtype=torch.float32).pin_memory()
            else:
                self._pinned = torch.empty((self.max_streams, self.chunk_samples), dtype=torch.float32).pin_memory()
        else:
            self.batch, self.rings = None, None
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
            if self.rings is not None:
                self.rings.reset_row(slot)
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
        pending for it.  The windowing is ``rearrange_audio_stream`` (``operators.py:44-100``): whole
        ``step`` blocks, a window once ``duration`` seconds have arrived, then one per block."""
        x = np.asarray(samples, dtype=np.float32).reshape(-1)
        with self._lock:
            st = self._streams[stream_id]
            st.buffer = np.concatenate([st.buffer, x]) if st.buffer.size else x.copy()
            while st.buffer.size >= self.step_samples:
                st.blocks.append(st.buffer[:self.step_samples].copy())
                st.buffer = st.buffer[self.step_samples:]
            return self._pending(st)

    def pending(self, stream_id: Hashable) -> int:
        """Windows of the stream still waiting for the worker (0 for an unknown / closed stream)."""
        with self._lock:
            st = self._streams.get(stream_id)
            return 0 if st is None else self._pending(st)

    def _pending(self, st: _Stream) -> int:
        # block number c (1-based) completes a window iff c >= blocks_per_window
        last = st.consumed + len(st.blocks)
        return max(0, last - max(st.consumed, self.blocks_per_window - 1))

    def _take_host_window(self, st: _Stream):
        """Host-window mode: consume blocks until one completes a window -> (window copy, start)."""
        while st.blocks:
            new = st.blocks.pop(0)
            st.consumed += 1
            st.chunk = new if st.chunk is None else np.concatenate([st.chunk, new])
            if st.chunk.size > self.chunk_samples:
                st.chunk = st.chunk[-self.chunk_samples:]
                st.start += self.step_seconds
            if st.chunk.size == self.chunk_samples:
                return st.chunk.copy(), st.start
        return None

    def _take_ring_blocks(self, st: _Stream):
        """Ring mode: the blocks to upload this step (until one completes a window) -> (blocks,
        start time of the completed window | None)."""
        taken = []
        while st.blocks:
            taken.append(st.blocks.pop(0))
            st.consumed += 1
            if st.consumed >= self.blocks_per_window:
                return taken, (st.consumed - self.blocks_per_window) * self.step_seconds
        return taken, None

    # ------------------------------------------------------------------ the worker
    def step(self) -> Dict[Hashable, Annotation]:
        """Process at most one pending window of every open stream, as ONE batch.  Returns the
        speech turns each of those streams gained (the per-chunk ``Annotation`` of the reference's
        pipeline); the running total is kept per stream until ``close``."""
        with self._step_lock:                       # one engine, one step at a time
            uploads = []                            # ring mode: (slot, [blocks]) of this step
            undo = []                               # (stream, blocks, consumed, chunk, start) before the take
            pushed: Dict[int, int] = {}             # ring mode: blocks of a slot that reached its ring
            with self._lock:
                work = []                           # (stream id, stream, host window | None, start)
                for sid, st in self._streams.items():
                    if not st.blocks:
                        continue
                    undo.append((st, list(st.blocks), st.consumed, st.chunk, st.start))
                    if self.rings is None:
                        got = self._take_host_window(st)
                        if got is not None:
                            work.append((sid, st, got[0], got[1]))
                    else:
                        taken, start = self._take_ring_blocks(st)
                        uploads.append((st.slot, taken))
                        if start is not None:
                            work.append((sid, st, None, start))
                # slots whose ring row / clustering state the worker touches until the step is over
                self._inflight = {st.slot for _, st, _, _ in work} | {slot for slot, _ in uploads}
            try:
                if uploads:
                    self._upload(uploads, pushed)
                if not work:
                    if uploads:     # warm-up blocks only: the staging planes are free once the GPU has read them
                        self._sync_uploads()
                    return {}
                starts = np.array([t for _, _, _, t in work], dtype=np.float64)
                slots = [st.slot for _, st, _, _ in work]
                if self.rings is None:
                    turns = self._engine(np.stack([w for _, _, w, _ in work]), starts, slots)
                else:
                    turns = self._gpu_engine(self.rings.gather(slots, self._dev), starts, slots)
            except BaseException as exc:
                self._roll_back(undo, pushed, exc)
                raise
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

    def _roll_back(self, undo, pushed: Dict[int, int], exc: BaseException) -> None:
        """A step failed (upload, gather or the engine raised).  The blocks it took must not vanish
        and the host-side block count of a stream must keep matching what its device ring holds:
        otherwise every later step of that stream computes wrong start times or fails in
        ``dz_ring_gather`` — which rejects the whole batch, i.e. one desynchronised stream would fail
        every step of every stream.  Host-window mode: the taken blocks go back to the
        front of the queue and the window state is restored; the step can simply be retried.  Ring
        mode: blocks that already reached the ring stay consumed (``consumed`` = count before + pushed,
        the ring's own per-row counter), the others go back; a window whose blocks were all pushed
        but whose engine run failed is lost for that stream (recorded in ``step_errors``)."""
        with self._lock:
            for st, blocks, consumed, chunk, start in undo:
                # blocks this step popped: both take functions count every pop in `consumed`.  (NOT len(blocks)
                # - len(st.blocks): a push() that lands while the engine runs — the lock is released then —
                # makes that short, and taken blocks would be dropped)  st.blocks now holds the
                # untaken blocks followed by whatever was pushed meanwhile.
                taken = st.consumed - consumed
                keep = pushed.get(st.slot, 0) if self.rings is not None else 0
                keep = min(keep, taken)
                st.blocks = blocks[keep:taken] + st.blocks
                st.consumed = consumed + keep
                if self.rings is None:
                    st.chunk, st.start = chunk, start
            self.step_errors.append((time.time(), repr(exc)))

    def _sync_uploads(self) -> None:
        if self._dev.is_cuda:
            torch.cuda.current_stream(self._dev.device).synchronize()

    def _upload(self, uploads, pushed: Optional[Dict[int, int]] = None) -> None:
        """New blocks -> the streams' device rings: round r carries the r-th new block of every stream
        that has one (a running stream has exactly one; a joining stream up to a whole window).
        ``pushed[slot]`` counts the blocks of a slot whose ring push was enqueued (for ``_roll_back``)."""
        import contextlib
        with (torch.cuda.device(self._dev.device) if self._dev.is_cuda else contextlib.nullcontext()):
            for r in range(max(len(b) for _, b in uploads)):
                rows = [slot for slot, b in uploads if len(b) > r]
                plane = self._stage[r]
                for j, (slot, b) in enumerate((u for u in uploads if len(u[1]) > r)):
                    plane[j].copy_(torch.from_numpy(b[r]))
                self.rings.push_rows(plane[:len(rows)], rows)
                if pushed is not None:
                    for slot in rows:
                        pushed[slot] = pushed.get(slot, 0) + 1

    def drain(self) -> int:
        """``step`` until no block is waiting; returns the number of steps."""
        n = 0
        while True:
            with self._lock:
                if not any(st.blocks for st in self._streams.values()):
                    return n
            self.step()
            n += 1

    def serve_forever(self, idle_sleep: float = 0.002) -> None:
        while not self._stop:
            with self._lock:
                idle = not any(st.blocks for st in self._streams.values())
            if idle:
                time.sleep(idle_sleep)
            else:
                self.step()

    def shutdown(self) -> None:
        self._stop = True

    # ------------------------------------------------------------------ GPU engine
    def _gpu_engine(self, windows, starts: np.ndarray, slots: List[int]):
        """``windows``: (k, S) on the host (uploaded whole, like the reference) or already a device
        batch gathered from the rings."""
        k = windows.shape[0]
        if not torch.is_tensor(windows):
            self._pinned[:k].copy_(torch.from_numpy(windows))
            self._dev[:k].copy_(self._pinned[:k], non_blocking=True)
        ticket = self.batch.launch(self._dev[:k], starts, slots=slots)
        self.batch.finish(ticket, want_scores=False)
        _, _, _, _, turns, nturns = ticket["tail"]
        return [turns[i, :int(nturns[i])].copy() for i in range(k)]
