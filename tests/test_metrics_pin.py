"""Pins for the DER scorer and the RTTM reader / writer: every DER gate of the
repository rests on ``diart_amd.metrics``; ``pyannote.metrics`` (what the reference uses,
``/root/reference/src/diart/blocks/diarization.py:131-133``) cannot be installed here, so the scorer
is cross-checked against an INDEPENDENT brute-force evaluation of the same definition: a 1 ms frame
grid and an exhaustive search over every one-to-one speaker mapping.
"""
import io
import itertools
from pathlib import Path

import numpy as np
import pytest

from diart_amd.features import Annotation, Segment, load_rttm
from diart_amd.metrics import DetectionErrorRate, DiarizationErrorRate

GOLDEN = Path(__file__).resolve().parent / "golden"


def random_annotation(rng, labels, total_ms, uri):
    ann = Annotation(uri=uri, modality="speaker")
    grid = np.zeros((total_ms, len(labels)), dtype=bool)
    for j, lab in enumerate(labels):
        t = int(rng.integers(0, 3000))
        n = 0
        while t < total_ms:
            dur = int(rng.integers(50, 4000))
            end = min(total_ms, t + dur)
            ann[Segment(t / 1000.0, end / 1000.0), (lab, n)] = lab
            grid[t:end, j] = True
            t = end + int(rng.integers(1, 5000))        # >= 1 ms gap: turns of one label never touch
            n += 1
    return ann, grid


def brute_force(ref_grid, hyp_grid):
    """Components in seconds on the 1 ms grid; confusion minimised over ALL one-to-one mappings."""
    nref, nhyp = ref_grid.sum(1), hyp_grid.sum(1)
    total = nref.sum() / 1000.0
    miss = np.maximum(0, nref - nhyp).sum() / 1000.0
    fa = np.maximum(0, nhyp - nref).sum() / 1000.0
    R, H = ref_grid.shape[1], hyp_grid.shape[1]
    cooc = hyp_grid.T.astype(np.int64) @ ref_grid.astype(np.int64)       # (H, R) ms of co-activity
    best = 0
    slots = list(range(R)) + [None] * H                                   # a hyp speaker may stay unmapped
    for assign in set(itertools.permutations(slots, H)):
        best = max(best, sum(cooc[h, r] for h, r in enumerate(assign) if r is not None))
    correct = best / 1000.0
    conf = np.minimum(nref, nhyp).sum() / 1000.0 - correct
    return {"total": total, "correct": correct, "missed detection": miss, "false alarm": fa, "confusion": conf}


@pytest.mark.parametrize("seed,R,H", [(0, 2, 2), (1, 3, 2), (2, 2, 4), (3, 4, 4), (4, 1, 3), (5, 4, 5), (6, 3, 3)])
def test_der_equals_brute_force_frame_grid(seed, R, H):
    rng = np.random.default_rng(seed)
    total_ms = 20000
    ref, rg = random_annotation(rng, [f"r{i}" for i in range(R)], total_ms, "f")
    hyp, hg = random_annotation(rng, [f"speaker{i}" for i in range(H)], total_ms, "f")
    if seed % 2:                      # correlate: the hypothesis repeats some reference turns
        for seg, trk, lab in list(ref.itertracks(yield_label=True))[::2]:
            j = int(lab[1:]) % H
            hyp[seg, ("copy", trk)] = f"speaker{j}"
            hg[int(round(seg.start * 1000)):int(round(seg.end * 1000)), j] = True
    want = brute_force(rg, hg)
    got = DiarizationErrorRate()(ref, hyp, detailed=True)
    for k, v in want.items():
        assert abs(got[k] - v) < 1e-6, (k, got[k], v)
    err = want["missed detection"] + want["false alarm"] + want["confusion"]
    assert abs(got["diarization error rate"] - err / want["total"]) < 1e-9
    det = DetectionErrorRate()(ref, hyp, detailed=True)
    rs, hs = rg.any(1), hg.any(1)
    assert abs(det["missed detection"] - (rs & ~hs).sum() / 1000.0) < 1e-6
    assert abs(det["false alarm"] - (hs & ~rs).sum() / 1000.0) < 1e-6
    assert abs(det["total"] - rs.sum() / 1000.0) < 1e-6


def test_known_answers():
    ref, hyp = Annotation("u"), Annotation("u")
    ref[Segment(0, 10), 0] = "A"
    ref[Segment(5, 15), 1] = "B"
    hyp[Segment(0, 15), 0] = "x"          # covers A then B: mapped to the longer co-occurrence
    d = DiarizationErrorRate()(ref, hyp, detailed=True)
    assert d["total"] == 20 and d["false alarm"] == 0
    assert d["missed detection"] == 5                      # [5,10): two reference speakers, one hypothesis
    assert d["confusion"] == 5 and d["correct"] == 10     # x <-> A (or B): 10 s correct, 5 s confused
    assert abs(d["diarization error rate"] - 0.5) < 1e-12


def test_rttm_reader_and_writer_on_the_reference_expected_output():
    """tests/golden/ami_0.5s_slice.rttm = lines of /root/reference/expected_outputs/online/0.5s/AMI.rttm
    (the paper implementation's output, labels A, B, ...): parse, re-emit, compare text and durations."""
    path = GOLDEN / "ami_0.5s_slice.rttm"
    anns = load_rttm(path)
    assert sorted(anns) == ["IS1009a", "IS1009b"]
    text = path.read_text().splitlines()
    for uri, ann in anns.items():
        lines = [ln for ln in text if ln.split()[1] == uri]
        assert len(ann) == len(lines) == 40
        assert set(ann.labels()) <= set("ABCDEFGH")
        want_total = sum(float(ln.split()[4]) for ln in lines)
        got_total = sum(seg.duration for seg, _ in ann.itertracks())
        assert abs(got_total - want_total) < 1e-6
        buf = io.StringIO()
        ann.write_rttm(buf)
        # same lines (the file is sorted by start time within a meeting, like itertracks)
        assert sorted(buf.getvalue().splitlines()) == sorted(lines)
    # a hypothesis scored against itself is perfect; against the other meeting it is not
    a, b = anns["IS1009a"], anns["IS1009b"]
    assert DiarizationErrorRate()(a, a) == 0.0
    assert DiarizationErrorRate()(a, b) > 0.1


def test_support_merges_only_gaps_shorter_than_the_collar():
    """pyannote.core Timeline.support: touching / overlapping turns merge, a gap merges iff its
    duration is < collar (strict) — what PredictionAccumulator's stitching relies on."""
    ann = Annotation("u")
    # binary fractions: the arithmetic of the gaps is exact
    for n, (s, e) in enumerate([(0.0, 1.0), (1.0, 2.0), (2.5, 3.0), (3.25, 4.0), (5.0, 6.0)]):
        ann[Segment(s, e), n] = "A"
    turns = [(s.start, s.end) for s, _ in ann.support(0.5).itertracks()]
    assert turns == [(0.0, 2.0), (2.5, 4.0), (5.0, 6.0)]      # gap 0.5 stays (not < collar), 0.25 merges
    turns0 = [(s.start, s.end) for s, _ in ann.support().itertracks()]
    assert turns0 == [(0.0, 2.0), (2.5, 3.0), (3.25, 4.0), (5.0, 6.0)]
