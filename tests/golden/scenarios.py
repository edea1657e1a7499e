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
