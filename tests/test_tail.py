"""Aggregation / binarisation tail (host code, no GPU): the product blocks and the oracle
restatement against ``tests/golden/tail.npz`` — outputs of the reference's own
``blocks/aggregation.py`` + ``blocks/utils.py`` (tests/golden/make_golden.py)."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import scenarios  # noqa: E402

GOLD = np.load(Path(__file__).resolve().parent / "golden" / "tail.npz")


def _drive(make_swf, make_pred, make_audio, make_mean, binarize, start_time, latency):
    scores, starts, res = scenarios.tail_inputs(start_time)
    pred, audio, mean = make_pred(latency), make_audio(latency), make_mean(latency)
    pbuf, abuf = [], []
    for i in range(scores.shape[0]):
        pbuf.append(make_swf(scores[i], starts[i], res))
        wav = (np.arange(80000, dtype=np.float64) + 8000.0 * i)[:, None]
        abuf.append(make_swf(wav, starts[i], 1 / 16000))
        agg, aud, mn = pred(pbuf), audio(abuf), mean(pbuf)
        yield i, agg, aud, mn, binarize(agg)
        if len(pbuf) == pred.num_overlapping_windows:
            pbuf, abuf = pbuf[1:], abuf[1:]


def _check(i, tag0, latency, agg, aud, mn, turns):
    tag = f"{tag0}_l{latency:g}_t{i}"
    assert agg.data.shape == GOLD[tag + "_agg"].shape
    assert np.array_equal(agg.data, GOLD[tag + "_agg"]), tag
    assert np.allclose([agg.sliding_window.start, agg.sliding_window.step], GOLD[tag + "_aggsw"], rtol=0, atol=1e-12)
    assert np.array_equal(mn.data, GOLD[tag + "_mean"]), tag
    want = GOLD[tag + "_turns"]
    got = np.array(turns, dtype=np.float64).reshape(-1, 3)
    assert got.shape == want.shape and np.allclose(got, want, rtol=0, atol=1e-12), tag
    a = GOLD[tag + "_aud"]
    assert aud.data.shape[0] == int(a[0]) and aud.data[0, 0] == a[1] and aud.data[-1, 0] == a[2]
    assert np.allclose([aud.sliding_window.start, aud.sliding_window.step], a[3:], rtol=0, atol=1e-12)


@pytest.mark.parametrize("start_time", [0.0, 7.5])
@pytest.mark.parametrize("latency", scenarios.TAIL_LATENCIES)
def test_product_tail_matches_reference_golden(start_time, latency):
    from diart_amd.blocks import Binarize, DelayedAggregation
    from diart_amd.features import SlidingWindow, SlidingWindowFeature
    binarize = Binarize(0.5)

    def turns(agg):
        return sorted((s.start, s.end, float(t)) for s, t, _ in binarize(agg).itertracks(yield_label=True))
    for i, agg, aud, mn, tr in _drive(
            lambda d, s, r: SlidingWindowFeature(d, SlidingWindow(start=s, duration=r, step=r)),
            lambda l: DelayedAggregation(0.5, l, "hamming", "loose"),
            lambda l: DelayedAggregation(0.5, l, "first", "center"),
            lambda l: DelayedAggregation(0.5, l, "mean", "strict"), turns, start_time, latency):
        _check(i, f"s{start_time:g}", latency, agg, aud, mn, tr)


@pytest.mark.parametrize("start_time", [0.0, 7.5])
@pytest.mark.parametrize("latency", scenarios.TAIL_LATENCIES)
def test_oracle_tail_matches_reference_golden(start_time, latency):
    from oracle.pyannote_stub import SlidingWindow, SlidingWindowFeature
    from oracle.tail_ref import DelayedAggregationRef, binarize_ref
    for i, agg, aud, mn, tr in _drive(
            lambda d, s, r: SlidingWindowFeature(d, SlidingWindow(start=s, duration=r, step=r)),
            lambda l: DelayedAggregationRef(0.5, l, "hamming", "loose"),
            lambda l: DelayedAggregationRef(0.5, l, "first", "center"),
            lambda l: DelayedAggregationRef(0.5, l, "mean", "strict"),
            lambda agg: binarize_ref(agg, 0.5), start_time, latency):
        _check(i, f"s{start_time:g}", latency, agg, aud, mn, tr)


@pytest.mark.parametrize("latency", scenarios.TAIL_LATENCIES)
def test_cpp_batched_tail_matches_reference_golden(latency):
    """dz_tail_step_batch (C++, what StreamBatch uses for N streams) against the same goldens of
    the reference's aggregation.py + utils.py: two streams (starting at 0 s and 7.5 s) stepped
    together; hamming/loose aggregation + turns, mean/strict aggregation, first/center audio."""
    from diart_amd.blocks.aggregation import BatchedOutputTail
    starts_of = (0.0, 7.5)
    inputs = [scenarios.tail_inputs(s0) for s0 in starts_of]
    T, F, G = inputs[0][0].shape
    res = inputs[0][2]
    pred = BatchedOutputTail(2, F, G, 0.5, latency, threshold=0.5, num_threads=2)
    mean = BatchedOutputTail(2, F, G, 0.5, latency, strategy="mean", cropping_mode="strict", num_threads=1)
    audio = BatchedOutputTail(2, 80000, 1, 0.5, latency, strategy="first", cropping_mode="center",
                              max_turns=4)
    assert pred.num_overlapping_windows == int(round(latency / 0.5))
    for i in range(T):
        scores = np.stack([inp[0][i] for inp in inputs])
        starts = np.array([inp[1][i] for inp in inputs])
        agg, rows, t0, r, turns, nturns = pred(scores, starts, res)
        magg, mrows, _, _, _, _ = mean(scores, starts, res)
        wav = np.repeat((np.arange(80000, dtype=np.float64) + 8000.0 * i)[None, :, None], 2, axis=0)
        aagg, arows, at0, ar, _, _ = audio(wav, starts, 1 / 16000)
        for k, s0 in enumerate(starts_of):
            tag = f"s{s0:g}_l{latency:g}_t{i}"
            want = GOLD[tag + "_agg"]
            assert rows[k] == want.shape[0] and np.array_equal(agg[k, :rows[k]], want), tag
            assert np.allclose([t0[k], r[k]], GOLD[tag + "_aggsw"], rtol=0, atol=1e-12), tag
            assert np.array_equal(magg[k, :mrows[k]], GOLD[tag + "_mean"]), tag
            wt = GOLD[tag + "_turns"]
            got = np.array(sorted(map(tuple, turns[k, :nturns[k]])), dtype=np.float64).reshape(-1, 3)
            assert got.shape == wt.shape and np.allclose(got, wt, rtol=0, atol=1e-12), tag
            a = GOLD[tag + "_aud"]
            assert arows[k] == int(a[0]) and aagg[k, 0, 0] == a[1] and aagg[k, arows[k] - 1, 0] == a[2], tag
            assert np.allclose([at0[k], ar[k]], a[3:], rtol=0, atol=1e-12), tag
    pred.reset()
    agg, rows, *_ = pred(np.stack([inp[0][0] for inp in inputs]), np.array([0.0, 7.5]), res)
    assert np.array_equal(agg[0, :rows[0]], GOLD[f"s0_l{latency:g}_t0_agg"])


def test_docstring_example_of_the_reference():
    """aggregation.py:141-161: 5 s / 500 frames / step 0.5 / latency 2 -> 4 windows, (51, 2)."""
    from diart_amd.blocks import DelayedAggregation
    from diart_amd.features import SlidingWindow, SlidingWindowFeature
    dagg = DelayedAggregation(step=0.5, latency=2, strategy="mean")
    res = 5 / 500
    bufs = [SlidingWindowFeature(np.random.rand(500, 2), SlidingWindow(start=(i + 10) * 0.5, duration=res, step=res))
            for i in range(dagg.num_overlapping_windows)]
    assert dagg.num_overlapping_windows == 4 and dagg(bufs).data.shape == (51, 2)


def test_der_metric_known_answers():
    from diart_amd.features import Annotation, Segment
    from diart_amd.metrics import DetectionErrorRate, DiarizationErrorRate
    ref, hyp = Annotation("f"), Annotation("f")
    ref[Segment(0, 10), 0] = "A"
    ref[Segment(5, 15), 1] = "B"          # 5 s of overlap: total = 20
    hyp[Segment(0, 10), 0] = "x"          # = A
    hyp[Segment(10, 15), 1] = "y"         # = B on [10,15]; B missed on [5,10]
    hyp[Segment(15, 17), 2] = "y"         # 2 s false alarm
    der = DiarizationErrorRate()
    d = der(ref, hyp, detailed=True)
    assert d["total"] == 20 and d["missed detection"] == 5 and d["false alarm"] == 2 and d["confusion"] == 0
    assert abs(d["diarization error rate"] - 7 / 20) < 1e-12 and abs(abs(der) - 7 / 20) < 1e-12
    # label permutation does not matter, swapping speakers mid-way is confusion
    hyp2 = Annotation("f")
    hyp2[Segment(0, 5), 0] = "p"
    hyp2[Segment(5, 10), 1] = "q"
    ref2 = Annotation("f")
    ref2[Segment(0, 10), 0] = "A"
    assert abs(DiarizationErrorRate()(ref2, hyp2) - 0.5) < 1e-12
    assert DiarizationErrorRate()(ref, ref) == 0.0
    det = DetectionErrorRate()(ref, hyp, detailed=True)
    assert det["total"] == 15 and det["false alarm"] == 2 and det["missed detection"] == 0


def test_rttm_roundtrip(tmp_path):
    from diart_amd.features import Annotation, Segment, load_rttm
    ann = Annotation("meeting1")
    ann[Segment(0.5, 2.25), 0] = "speaker0"
    ann[Segment(1.0, 3.0), 1] = "speaker1"
    p = tmp_path / "x.rttm"
    with open(p, "w") as f:
        ann.write_rttm(f)
    assert p.read_text().splitlines()[0] == "SPEAKER meeting1 1 0.500 1.750 <NA> <NA> speaker0 <NA> <NA>"
    back = load_rttm(p)["meeting1"]
    assert [(s.start, s.end, l) for s, _, l in back.itertracks(yield_label=True)] == \
           [(0.5, 2.25, "speaker0"), (1.0, 3.0, "speaker1")]
