"""Seeded inputs for the clustering / functional fixtures (shared by make_golden.py and tests)."""
import numpy as np

# name -> (seed, steps, frames, local speakers K, dim D, max_speakers G, tau, rho, delta, pool, noise)
CLUSTERING = {
    "default":   (1, 80, 32, 3, 24, 20, 0.6, 0.3, 1.0, 6, 0.35),
    "ami_tuned": (2, 80, 32, 4, 24, 20, 0.507, 0.006, 1.057, 7, 0.5),
    "crowded":   (3, 90, 16, 3, 16, 4, 0.5, 0.2, 0.7, 9, 0.3),   # more speakers than centroids
    "tight":     (4, 60, 16, 3, 8, 20, 0.6, 0.3, 0.25, 4, 0.6),   # many near-threshold distances
}


def clustering_inputs(name):
    seed, T, F, K, D, G, tau, rho, delta, pool, noise = CLUSTERING[name]
    rng = np.random.default_rng(seed)
    voices = rng.standard_normal((pool, D))
    voices /= np.linalg.norm(voices, axis=1, keepdims=True)
    seg = np.zeros((T, F, K), dtype=np.float32)
    emb = np.zeros((T, K, D), dtype=np.float32)
    for t in range(T):
        who = rng.choice(pool, size=K, replace=False)
        for k in range(K):
            mode = rng.random()
            if t == 0 and name == "default":
                mode = 0.99 if k else 0.0          # first call: one silent, others active
            if mode < 0.2:      # silent speaker
                level, dens = 0.3 * rng.random(), 0.2
            elif mode < 0.35:   # short burst: active but not "long"
                level, dens = 0.7 + 0.3 * rng.random(), 0.08
            else:               # long turn
                level, dens = 0.6 + 0.4 * rng.random(), 0.5 + 0.5 * rng.random()
            act = (rng.random(F) < dens).astype(np.float32)
            seg[t, :, k] = np.clip(act * level + 0.05 * rng.random(F), 0, 1)
            e = voices[who[k]] + noise * rng.standard_normal(D) / np.sqrt(D)
            emb[t, k] = (e / np.linalg.norm(e)).astype(np.float32)
        r = rng.random()
        if r < 0.08:
            emb[t, rng.integers(K)] = np.nan       # ECAPA too-short / zero-weight speaker
        elif r < 0.12:
            seg[t] = 0.0                            # nobody speaks
        elif r < 0.16:
            emb[t, 1] = emb[t, 0]                   # identical embeddings -> tied distances
    return dict(seg=seg, emb=emb, tau=tau, rho=rho, delta=delta, G=G)


def functional_inputs():
    rng = np.random.default_rng(123)
    seg = rng.random((4, 293, 3)).astype(np.float32)
    seg[1, :, 0] = 0.0
    seg[2] = 0.5
    seg[3, :40] = 1.0
    emb = rng.standard_normal((4, 3, 64)).astype(np.float32)
    emb[1, 2] = 0.0   # zero embedding -> NaN after normalisation (functional.py:26-27)
    return seg, emb


# ---- aggregation / binarize tail (blocks/aggregation.py, blocks/utils.py) ---------------------
TAIL_LATENCIES = (0.5, 1.0, 2.5, 5.0)
TAIL_STEPS, TAIL_FRAMES, TAIL_SPEAKERS, TAIL_STEP, TAIL_DURATION = 16, 293, 5, 0.5, 5.0


def tail_inputs(start_time: float = 0.0):
    """Permuted score windows of one stream: smooth per-speaker activity that persists across the
    overlapping windows (so the Hamming average is not trivially noise), some all-zero columns
    (unassigned global speakers), values exactly at the threshold, windows starting at
    ``start_time + 0.5 i``."""
    rng = np.random.default_rng(77)
    res = TAIL_DURATION / TAIL_FRAMES
    total = int(round((TAIL_DURATION + TAIL_STEP * TAIL_STEPS) / res)) + 8
    track = np.zeros((total, TAIL_SPEAKERS))
    for k in range(TAIL_SPEAKERS - 1):
        state, t = rng.random() < 0.5, 0
        while t < total:
            n = int(rng.integers(20, 140))
            track[t:t + n, k] = (0.7 + 0.3 * rng.random()) if state else 0.1 * rng.random()
            state, t = not state, t + n
    scores = np.zeros((TAIL_STEPS, TAIL_FRAMES, TAIL_SPEAKERS))
    for i in range(TAIL_STEPS):
        off = int(round(i * TAIL_STEP / res))
        scores[i] = np.clip(track[off:off + TAIL_FRAMES] + 0.08 * rng.standard_normal((TAIL_FRAMES, TAIL_SPEAKERS)), 0, 1)
        scores[i][:, TAIL_SPEAKERS - 1] = 0.0
    scores[3, 100:110, 0] = 0.5            # exactly tau: Binarize is strict (>)
    starts = start_time + TAIL_STEP * np.arange(TAIL_STEPS)
    return scores, starts, res


# ---- whole pipelines (blocks/diarization.py, blocks/vad.py) around toy models -------------------
# Deterministic stand-ins with the surface the reference's LazyModel expects of a loaded model and
# that HipSegmentation / HipEmbedding offer: ``__call__`` + ``.to(device)``, not an nn.Module
# (README.md:186-209, models.py:112-139).  Plain torch arithmetic, device agnostic.
PIPE_SR, PIPE_SECONDS, PIPE_FRAMES, PIPE_DIM = 16000, 24.0, 100, 16
PIPE_CASES = {   # name -> (latency, batch size, timestamp shift, tau, rho, delta)
    "lat0.5_b4":       (0.5, 4, 0.0, 0.6, 0.3, 1.0),
    "lat2.0_b7_shift": (2.0, 7, -1.25, 0.55, 0.2, 0.8),
    "lat5.0_b1":       (5.0, 1, 0.0, 0.6, 0.3, 1.0),
}


def pipeline_stream():
    """24 s, three "speakers" = sinusoids at 300 / 1100 / 2700 Hz taking turns (some overlap, some
    silence), float32 in [-1, 1]."""
    rng = np.random.default_rng(2025)
    n = int(PIPE_SR * PIPE_SECONDS)
    t = np.arange(n) / PIPE_SR
    x = np.zeros(n)
    for k, f in enumerate((300.0, 1100.0, 2700.0)):
        on, pos, state = np.zeros(n), 0, rng.random() < 0.5
        while pos < n:
            length = int(PIPE_SR * (0.6 + 2.4 * rng.random()))
            if state:
                on[pos:pos + length] = 0.25 + 0.1 * rng.random()
            state, pos = not state, pos + length
        x += on * np.sin(2 * np.pi * f * t + k)
    x += 0.01 * rng.standard_normal(n)
    return np.clip(x, -1, 1).astype(np.float32)


def _toy_frames(wave):
    """(B, 1, S) -> per-frame energy of three band-pass-ish projections, (B, F, 3)."""
    import torch
    B, _, S = wave.shape
    L = S // PIPE_FRAMES
    fr = wave[:, 0, :L * PIPE_FRAMES].reshape(B, PIPE_FRAMES, L)
    tt = torch.arange(L, dtype=wave.dtype, device=wave.device) / PIPE_SR
    out = []
    for f in (300.0, 1100.0, 2700.0):
        c, s = torch.cos(2 * np.pi * f * tt), torch.sin(2 * np.pi * f * tt)
        out.append(torch.sqrt((fr * c).mean(-1) ** 2 + (fr * s).mean(-1) ** 2))
    return torch.stack(out, -1)


class ToySegmentation:
    def to(self, device):
        return self

    def __call__(self, waveform):
        import torch
        return torch.sigmoid(120.0 * (_toy_frames(waveform) - 0.05))          # (B, F, 3) in (0, 1)


class ToyEmbedding:
    def __init__(self):
        self.proj = None

    def to(self, device):
        return self

    def __call__(self, waveform, weights=None):
        import torch
        e = _toy_frames(waveform)                                              # (N, F, 3)
        if self.proj is None or self.proj.device != e.device:
            g = np.random.default_rng(99).standard_normal((3, PIPE_DIM)).astype(np.float32)
            self.proj = torch.from_numpy(g).to(e.device)
        if weights is None:
            pooled = e.mean(1)
        else:
            w = weights.to(e.dtype)
            pooled = (e * w[..., None]).sum(1) / w.sum(1, keepdim=True)
        return torch.tanh(8.0 * pooled) @ self.proj                           # (N, D)


# ---- long-horizon parity (tests/test_gpu_long_horizon.py, make_long_horizon.py) -------------------------------------
LONG_SECONDS = 600.0
LONG_STREAMS = ((7001, 3), (7002, 4), (7003, 5), (7004, 3))          # (seed, speakers taking turns)
LONG_LATENCIES = (0.5, 5.0)
LONG_CENTER_STEPS = (50, 150, 400, 800, 1190)                       # chunk indices at which centroids are kept


def long_horizon_audio(i: int):
    """Stream i of the long-horizon scenario as a 16-bit WAV holds it (write_wav -> read_wav of
    diart_amd/inference.py): float32 samples k / 32768."""
    import numpy as np
    from diart_amd.synth import synth_stream
    seed, nspk = LONG_STREAMS[i]
    x = synth_stream(seed, LONG_SECONDS, num_speakers=nspk)
    pcm = np.clip(np.rint(np.asarray(x, dtype=np.float64) * 32768.0), -32768, 32767).astype("<i2")
    return pcm.astype(np.float32) / 32768.0
