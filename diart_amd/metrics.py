"""Diarization / detection error rates (``pyannote.metrics`` is not installable here).

``DiarizationErrorRate(collar=0, skip_overlap=False)`` is what the reference's pipelines suggest
(``/root/reference/src/diart/blocks/diarization.py:131-133``; ``vad.py:108-110`` for
``DetectionErrorRate``) and what ``Benchmark`` reports (``inference.py:380-389``).  Definition
(NIST md-eval, as pyannote.metrics implements it): with the one-to-one speaker mapping that
maximises the total overlap (Hungarian), over every elementary region with ``Nref`` reference and
``Nhyp`` hypothesis speakers of which ``Ncor`` are correctly mapped,

    miss = max(0, Nref - Nhyp), false alarm = max(0, Nhyp - Nref),
    confusion = min(Nref, Nhyp) - Ncor,   DER = sum(miss + fa + conf) / sum(Nref)   (durations)

No collar, overlapped speech included.  Components accumulate over files like BaseMetric.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .features import Annotation

COMPONENTS = ("total", "correct", "false alarm", "missed detection", "confusion")


def _turns(ann: Annotation) -> List[Tuple[float, float, object]]:
    return [(seg.start, seg.end, label) for seg, _, label in ann.support().itertracks(yield_label=True)]


def _regions(ref_turns, hyp_turns):
    """Elementary regions between consecutive boundaries -> (duration, ref labels, hyp labels)."""
    bounds = sorted({t for s, e, _ in ref_turns + hyp_turns for t in (s, e)})
    events = []  # (time, +1/-1, side, label)
    for side, turns in ((0, ref_turns), (1, hyp_turns)):
        for s, e, label in turns:
            events.append((s, 1, side, label))
            events.append((e, -1, side, label))
    events.sort(key=lambda ev: (ev[0], ev[1]))
    active = ({}, {})
    k = 0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        while k < len(events) and events[k][0] <= lo:
            _, d, side, label = events[k]
            active[side][label] = active[side].get(label, 0) + d
            k += 1
        r = [lab for lab, c in active[0].items() if c > 0]
        h = [lab for lab, c in active[1].items() if c > 0]
        if r or h:
            yield hi - lo, r, h


def optimal_mapping(reference: Annotation, hypothesis: Annotation) -> Dict[object, object]:
    """hypothesis label -> reference label maximising the co-occurrence duration."""
    from ._lsap import linear_sum_assignment
    rl, hl = reference.labels(), hypothesis.labels()
    if not rl or not hl:
        return {}
    cooc = np.zeros((len(hl), len(rl)))
    ri, hi = {l: i for i, l in enumerate(rl)}, {l: i for i, l in enumerate(hl)}
    for dur, r, h in _regions(_turns(reference), _turns(hypothesis)):
        for a in h:
            for b in r:
                cooc[hi[a], ri[b]] += dur
    rows, cols = linear_sum_assignment(-cooc)
    return {hl[i]: rl[j] for i, j in zip(rows, cols) if cooc[i, j] > 0}


class _Accumulating:
    def __init__(self):
        self.accumulated = {c: 0.0 for c in COMPONENTS}
        self.results: List[Tuple[object, Dict[str, float]]] = []

    def _rate(self, comp: Dict[str, float]) -> float:
        err = comp["false alarm"] + comp["missed detection"] + comp["confusion"]
        return err / comp["total"] if comp["total"] > 0 else (0.0 if err == 0 else 1.0)

    def __call__(self, reference: Annotation, hypothesis: Annotation, detailed: bool = False):
        comp = self.components(reference, hypothesis)
        for c in COMPONENTS:
            self.accumulated[c] += comp[c]
        self.results.append((reference.uri, comp))
        rate = self._rate(comp)
        return dict(comp, **{self.name: rate}) if detailed else rate

    def __abs__(self) -> float:
        return self._rate(self.accumulated)

    def reset(self):
        self.__init__()

    def report(self) -> str:
        head = f"{'item':24s} {self.name + ' %':>8s} " + " ".join(f"{c:>17s}" for c in COMPONENTS)
        lines = [head]
        for uri, comp in self.results + [("TOTAL", self.accumulated)]:
            lines.append(f"{str(uri):24s} {100 * self._rate(comp):8.2f} " +
                         " ".join(f"{comp[c]:17.2f}" for c in COMPONENTS))
        return "\n".join(lines)


class DiarizationErrorRate(_Accumulating):
    name = "diarization error rate"

    def __init__(self, collar: float = 0.0, skip_overlap: bool = False):
        if collar != 0.0 or skip_overlap:
            raise NotImplementedError("only collar=0, skip_overlap=False (what diart suggests) is built")
        super().__init__()

    def components(self, reference: Annotation, hypothesis: Annotation) -> Dict[str, float]:
        mapping = optimal_mapping(reference, hypothesis)
        comp = {c: 0.0 for c in COMPONENTS}
        for dur, r, h in _regions(_turns(reference), _turns(hypothesis)):
            nref, nhyp = len(r), len(h)
            ncor = len(set(r) & {mapping.get(a) for a in h})
            comp["total"] += dur * nref
            comp["correct"] += dur * ncor
            comp["missed detection"] += dur * max(0, nref - nhyp)
            comp["false alarm"] += dur * max(0, nhyp - nref)
            comp["confusion"] += dur * (min(nref, nhyp) - ncor)
        return comp


class DetectionErrorRate(_Accumulating):
    name = "detection error rate"

    def __init__(self, collar: float = 0.0, skip_overlap: bool = False):
        if collar != 0.0 or skip_overlap:
            raise NotImplementedError("only collar=0, skip_overlap=False is built")
        super().__init__()

    def components(self, reference: Annotation, hypothesis: Annotation) -> Dict[str, float]:
        comp = {c: 0.0 for c in COMPONENTS}
        for dur, r, h in _regions(_turns(reference), _turns(hypothesis)):
            comp["total"] += dur * bool(r)
            comp["correct"] += dur * (bool(r) and bool(h))
            comp["missed detection"] += dur * (bool(r) and not h)
            comp["false alarm"] += dur * (bool(h) and not r)
        return comp
