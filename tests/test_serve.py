"""Host logic of the cross-stream server (diart_amd/serve.py) with a recording engine in place of
the GPU: per-stream windowing == rearrange_audio_stream for arbitrary block sizes, one window per
stream per step batched across streams, join / leave at any time, slot reuse, thread safety."""
import threading

import numpy as np
import pytest

from diart_amd.inference import file_blocks, rolling_windows
from diart_amd.serve import StreamServer


class Recorder:
    """engine(windows, starts, slots): remembers every batch; one turn per window =
    [start, start + 0.5) on 'speaker' = slot, if the window's newest sample is > 0."""

    def __init__(self):
        self.batches, self.resets = [], []

    def reset(self, slot):
        self.resets.append(slot)

    def __call__(self, windows, starts, slots):
        self.batches.append((windows.copy(), np.array(starts), list(slots)))
        return [np.array([[s + 4.5, s + 5.0, float(slot)]]) if w[-1] > 0 else np.zeros((0, 3))
                for w, s, slot in zip(windows, starts, slots)]


def ramp(n, offset=0):
    return (np.arange(n, dtype=np.float32) + offset + 1) / 1e6


def test_windows_follow_rearrange_audio_stream_for_any_block_size():
    rec = Recorder()
    srv = StreamServer(None, None, max_streams=4, engine=rec)
    audio = ramp(16000 * 9 + 1234)
    srv.open("a")
    pos, sizes = 0, [3000, 1, 7999, 16000, 40000, 123]
    i = 0
    while pos < len(audio):
        n = sizes[i % len(sizes)]
        srv.push("a", audio[pos:pos + n])
        pos += n
        i += 1
        srv.drain()
    want = list(rolling_windows(file_blocks(audio[:len(audio) // 8000 * 8000], 16000, (0, 0), 0.5), 5.0, 0.5, 16000))
    got = [(w[0], s[0]) for w, s, _ in rec.batches]
    assert len(got) == len(want) == (len(audio) - 80000) // 8000 + 1
    for (w, s), ref in zip(got, want):
        assert np.array_equal(w, ref.data[:, 0]) and abs(s - ref.sliding_window.start) < 1e-9
    total = srv.close("a")
    assert total.uri == "a" and len(total) == 1          # consecutive half-second turns are stitched
    seg = next(iter(total.itertracks()))[0]
    assert abs(seg.start - 4.5) < 1e-9 and abs(seg.end - (4.5 + 0.5 * len(want))) < 1e-9


def test_batches_across_streams_and_join_leave():
    rec = Recorder()
    srv = StreamServer(None, None, max_streams=2, engine=rec)
    srv.open("x")
    srv.open("y")
    with pytest.raises(RuntimeError):
        srv.open("z")                                     # no free slot
    with pytest.raises(ValueError):
        srv.open("x")
    assert rec.resets == [0, 1]
    srv.push("x", ramp(80000 + 16000))                    # 3 windows pending
    srv.push("y", ramp(80000, offset=5_000_000))          # 1 window pending
    assert srv.step().keys() == {"x", "y"}                # one window of each, ONE batch
    assert rec.batches[-1][0].shape == (2, 80000) and rec.batches[-1][2] == [0, 1]
    assert srv.step().keys() == {"x"} and srv.step().keys() == {"x"} and srv.step() == {}
    assert [b[2] for b in rec.batches] == [[0, 1], [0], [0]]
    assert list(rec.batches[1][1]) == [0.5] and list(rec.batches[2][1]) == [1.0]
    done = srv.close("y")
    assert done.uri == "y" and srv.open_streams == ["x"]
    srv.open("z")                                         # takes y's slot, state reset
    assert rec.resets[-1] == 1
    srv.push("z", ramp(80000))
    srv.push("x", ramp(8000))
    out = srv.step()
    assert set(out) == {"x", "z"} and rec.batches[-1][2] == [0, 1]
    assert list(rec.batches[-1][1]) == [1.5, 0.0]         # x is at its 4th window, z at its first
    with pytest.raises(KeyError):
        srv.push("y", ramp(10))


def test_push_is_thread_safe():
    rec = Recorder()
    srv = StreamServer(None, None, max_streams=8, engine=rec)
    ids = [f"s{i}" for i in range(8)]
    for sid in ids:
        srv.open(sid)

    def client(sid, seed):
        rng = np.random.default_rng(seed)
        audio, pos = ramp(80000 + 8000 * 6, offset=seed * 1000), 0
        while pos < len(audio):
            n = int(rng.integers(100, 9000))
            srv.push(sid, audio[pos:pos + n])
            pos += n

    worker = threading.Thread(target=srv.serve_forever, kwargs={"idle_sleep": 0.0005})
    worker.start()
    clients = [threading.Thread(target=client, args=(sid, i)) for i, sid in enumerate(ids)]
    for c in clients:
        c.start()
    for c in clients:
        c.join()
    srv.shutdown()
    worker.join(timeout=10)
    srv.drain()
    per_stream = {}
    for w, s, slots in rec.batches:
        for row, start, slot in zip(w, s, slots):
            per_stream.setdefault(slot, []).append((start, row[0]))
    assert len(per_stream) == 8
    for slot, seq in per_stream.items():
        assert [t for t, _ in seq] == [0.5 * i for i in range(7)]     # every window, in order
        first = [v for _, v in seq]
        assert np.allclose(np.diff(first), 8000 / 1e6, atol=1e-9)     # each window starts 8000 samples later


def test_close_during_a_step_defers_the_slot_and_drops_the_result():
    """close() while the worker is inside step() must not hand the slot to the next
    open() (whose reset would clear clustering / tail state that host threads are still reading);
    the in-flight result of the closed stream is discarded."""
    gate_in, gate_out = threading.Event(), threading.Event()

    class Slow(Recorder):
        def __call__(self, windows, starts, slots):
            gate_in.set()
            assert gate_out.wait(10)
            return super().__call__(windows, starts, slots)

    rec = Slow()
    srv = StreamServer(None, None, max_streams=2, engine=rec)
    srv.open("a")
    srv.open("b")
    srv.push("a", ramp(80000))
    srv.push("b", ramp(80000))
    result = {}
    worker = threading.Thread(target=lambda: result.update(srv.step()))
    worker.start()
    assert gate_in.wait(10)                     # the engine is now working on slots 0 and 1
    srv.close("a")                              # slot 0 is in flight: must not become free yet
    with pytest.raises(RuntimeError):
        srv.open("c")                           # ... so there is no slot for a newcomer
    assert rec.resets == [0, 1]                 # and nothing was reset under the worker
    gate_out.set()
    worker.join(timeout=10)
    assert set(result) == {"b"}                 # a's window was processed but its result dropped
    srv.open("c")                               # the deferred slot is free now
    assert rec.resets == [0, 1, 0]


def test_ring_mode_block_schedule_reproduces_the_host_windows():
    """The device-ring path of ``step`` uploads blocks instead of windows (``_take_ring_blocks``):
    replaying its (blocks, start) schedule into a host-side rolling buffer must give exactly the
    windows and start times of the host-window path, for any push pattern — including a joining
    stream that delivers several windows' worth of audio before the first step."""
    rec = Recorder()
    host = StreamServer(None, None, max_streams=1, engine=rec)
    ringy = StreamServer(None, None, max_streams=1, engine=Recorder())
    audio = ramp(16000 * 8 + 777)
    host.open("a")
    ringy.open("a")
    st = ringy._streams["a"]
    rolled, replay = np.zeros(0, dtype=np.float32), []
    pos, sizes, i = 0, [50000, 9000, 40000, 123, 8000, 8000, 31000], 0
    while pos < len(audio):
        n = sizes[i % len(sizes)]
        assert host.push("a", audio[pos:pos + n]) == ringy.push("a", audio[pos:pos + n])
        pos += n
        i += 1
        host.step()                                   # ONE step per push: windows queue up behind it
        taken, start = ringy._take_ring_blocks(st)
        assert len(taken) <= ringy.blocks_per_window
        for b in taken:
            rolled = np.concatenate([rolled, b])[-ringy.chunk_samples:]
        if start is not None:
            replay.append((rolled.copy(), start))
    host.drain()
    while st.blocks:
        taken, start = ringy._take_ring_blocks(st)
        for b in taken:
            rolled = np.concatenate([rolled, b])[-ringy.chunk_samples:]
        if start is not None:
            replay.append((rolled.copy(), start))
    assert len(replay) == len(rec.batches) > 5
    for (w, s), (hw, hs, _) in zip(replay, rec.batches):
        assert np.array_equal(w, hw[0]) and abs(s - hs[0]) < 1e-9


class FakeRings:
    """Host stand-in for ``AudioRing``'s per-row interface (push_rows / gather / reset_row): the
    server's ring-mode control flow can then run without a GPU."""

    def __init__(self, n, window, hop):
        self.window, self.hop = window, hop
        self.rows = [np.zeros(0, dtype=np.float32) for _ in range(n)]
        self.pushed_blocks = 0

    def push_rows(self, block, rows):
        assert tuple(block.shape) == (len(rows), self.hop) and len(set(rows)) == len(rows)
        for j, r in enumerate(rows):
            self.rows[r] = np.concatenate([self.rows[r], block[j].numpy()])[-self.window:]
        self.pushed_blocks += len(rows)

    def gather(self, rows, out):
        import torch
        for j, r in enumerate(rows):
            assert len(self.rows[r]) == self.window, "gather of an incomplete window"
            out[j].copy_(torch.from_numpy(self.rows[r]))
        return out[:len(rows)]

    def reset_row(self, r):
        self.rows[r] = np.zeros(0, dtype=np.float32)


def _ring_server(max_streams, rec):
    """A StreamServer whose ring-mode ``step`` runs on the host: fake rings, the recording engine
    behind ``_gpu_engine``."""
    import torch
    srv = StreamServer(None, None, max_streams=max_streams, engine=rec)
    srv.rings = FakeRings(max_streams, srv.chunk_samples, srv.step_samples)
    srv._stage = torch.empty((srv.blocks_per_window, max_streams, srv.step_samples), dtype=torch.float32)
    srv._dev = torch.empty((max_streams, srv.chunk_samples), dtype=torch.float32)
    srv._gpu_engine = lambda windows, starts, slots: rec(windows.numpy(), starts, slots)
    return srv


def test_ring_mode_step_equals_host_window_mode_for_streams_that_join_and_leave():
    rng = np.random.default_rng(5)
    rec_h, rec_r = Recorder(), Recorder()
    host = StreamServer(None, None, max_streams=3, engine=rec_h)
    ring = _ring_server(3, rec_r)
    audio = {k: ramp(int(16000 * d), offset=1000 * i) * (1 if i % 2 == 0 else -1)
             for i, (k, d) in enumerate({"a": 9.3, "b": 7.1, "c": 12.0, "d": 6.4}.items())}
    pos = {k: 0 for k in audio}
    join = {"a": 0, "b": 2, "c": 2, "d": 11}
    closed, tick, outs_h, outs_r = set(), 0, [], []
    while len(closed) < len(audio):
        for k in audio:
            if tick == join[k]:
                host.open(k)
                ring.open(k)
            if tick >= join[k] and k not in closed:
                n = int(rng.integers(1000, 40000))
                blk = audio[k][pos[k]:pos[k] + n]
                pos[k] += n
                if len(blk):
                    assert host.push(k, blk) == ring.push(k, blk)
                if pos[k] >= len(audio[k]) and not host._pending(host._streams[k]):
                    th, tr = host.close(k), ring.close(k)      # a slot is handed to "d" later
                    assert th.to_rttm() == tr.to_rttm()
                    closed.add(k)
        oh, orr = host.step(), ring.step()
        outs_h.append({k: v.to_rttm() for k, v in oh.items()})
        outs_r.append({k: v.to_rttm() for k, v in orr.items()})
        tick += 1
        assert tick < 500
    assert outs_h == outs_r and any(len(o) >= 2 for o in outs_r)
    # the engine saw the same windows, start times and slots, batch by batch
    assert len(rec_h.batches) == len(rec_r.batches) > 10
    for (wh, sh, lh), (wr, sr, lr) in zip(rec_h.batches, rec_r.batches):
        assert np.array_equal(wh, wr) and np.allclose(sh, sr) and lh == lr
    # only new blocks crossed the "bus": one 8000-sample block per consumed step block
    assert ring.rings.pushed_blocks == sum(len(a) // 8000 for a in audio.values())


class FlakyEngine(Recorder):
    """Raises on the calls listed in ``fail_on`` (0-based), records the others."""

    def __init__(self, fail_on):
        super().__init__()
        self.fail_on, self.calls = set(fail_on), 0

    def __call__(self, windows, starts, slots):
        i = self.calls
        self.calls += 1
        if i in self.fail_on:
            raise RuntimeError("engine down")
        return super().__call__(windows, starts, slots)


def test_failed_step_in_host_window_mode_can_be_retried_without_losing_a_window():
    """step() takes blocks before the engine runs; if the engine raises, the
    blocks go back and the window state is restored, so the retry sees the same window."""
    good = Recorder()
    ref = StreamServer(None, None, max_streams=2, engine=good)
    flaky = FlakyEngine(fail_on={1, 4})
    srv = StreamServer(None, None, max_streams=2, engine=flaky)
    audio = {"a": ramp(16000 * 8), "b": -ramp(16000 * 7, 500)}
    for s in (ref, srv):
        for k, x in audio.items():
            s.open(k)
            s.push(k, x)
    ref.drain()
    failures = 0
    for _ in range(100):
        try:
            if not srv.step() and not any(st.blocks for st in srv._streams.values()):
                break
        except RuntimeError:
            failures += 1
    assert failures == 2 and len(srv.step_errors) == 2
    assert len(flaky.batches) == len(good.batches)
    for (w, s, l), (rw, rs, rl) in zip(flaky.batches, good.batches):
        assert np.array_equal(w, rw) and np.allclose(s, rs) and l == rl
    for k in audio:
        assert srv.close(k).to_rttm() == ref.close(k).to_rttm()


def test_push_during_a_failing_step_does_not_lose_taken_blocks():
    """ADVICE r3 (serve.py _roll_back): the lock is released while the engine runs, so the I/O thread may
    push() then; if that step fails, the blocks it took go back IN FRONT of the newly pushed one, none lost."""
    srv = None

    class PushThenFail(Recorder):
        def __init__(self):
            super().__init__()
            self.calls = 0

        def __call__(self, windows, starts, slots):
            self.calls += 1
            if self.calls == 1:
                srv.push("a", np.full(8000, 9.0, dtype=np.float32))     # lands mid-step
                raise RuntimeError("engine down")
            return super().__call__(windows, starts, slots)

    eng = PushThenFail()
    srv = StreamServer(None, None, max_streams=1, engine=eng)
    srv.open("a")
    blocks = [np.full(8000, float(i + 1), dtype=np.float32) for i in range(11)]
    for b in blocks:
        srv.push("a", b)
    with pytest.raises(RuntimeError):
        srv.step()                       # takes 10 blocks (one window), the engine pushes block "9.0" and raises
    st = srv._streams["a"]
    assert st.consumed == 0
    assert [float(b[0]) for b in st.blocks] == [float(i + 1) for i in range(11)] + [9.0]
    srv.drain()
    # windows of the retried run: blocks 1..10, 2..11, 3..11+9
    assert [float(w[0][0]) for w, _, _ in eng.batches] == [1.0, 2.0, 3.0]
    assert float(eng.batches[-1][0][0][-1]) == 9.0


def test_failed_step_in_ring_mode_keeps_host_block_count_equal_to_the_ring():
    """Ring mode: blocks that reached the ring before the failure stay consumed (the ring cannot be
    rewound), blocks that did not go back to the queue — the stream's later windows and start times
    are the ones of an undisturbed run, and the other stream in the batch is not poisoned."""
    good, flaky = Recorder(), FlakyEngine(fail_on={2})
    ref, srv = _ring_server(2, good), _ring_server(2, flaky)
    # ring push that fails once, before anything of that round is written
    real_push, state = srv.rings.push_rows, {"n": 0}

    def push_rows(block, rows):
        state["n"] += 1
        if state["n"] == 3:
            raise OSError("bus error")
        return real_push(block, rows)

    srv.rings.push_rows = push_rows
    audio = {"a": ramp(16000 * 8), "b": -ramp(16000 * 9, 500)}
    for s in (ref, srv):
        for k in audio:
            s.open(k)
    pos = 0
    while pos < 16000 * 9:
        for s in (ref, srv):
            for k, x in audio.items():
                if pos < len(x):
                    s.push(k, x[pos:pos + 8000])
        pos += 8000
        ref.step()
        try:
            srv.step()
        except (RuntimeError, OSError):
            pass
    ref.drain()
    for _ in range(50):
        try:
            if not srv.drain():
                break
        except (RuntimeError, OSError):
            pass
    for st in srv._streams.values():      # host count == what the ring received, per stream
        assert not st.blocks
    assert srv.rings.pushed_blocks == ref.rings.pushed_blocks
    assert len(srv.step_errors) == 2
    # every window the flaky server did process is a window of the undisturbed run, same start
    ref_by_start = {}
    for w, s, l in good.batches:
        for wi, si, li in zip(w, s, l):
            ref_by_start[(li, round(float(si), 6))] = wi
    seen = 0
    for w, s, l in flaky.batches:
        for wi, si, li in zip(w, s, l):
            assert np.array_equal(wi, ref_by_start[(li, round(float(si), 6))])
            seen += 1
    total = sum(len(w) for w, _, _ in good.batches)
    assert total - 2 <= seen < total            # only the failed engine step's windows are missing
