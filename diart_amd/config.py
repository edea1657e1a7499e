"""Engine parameters.

Every knob of the engine is a constructor argument (``StreamBatch(recurrence=, lanes=, inflight=, wait=, warmup=)``,
``HipSegmentation(precision=, recurrence=)``, ``Benchmark(concurrent_files=)``, ...).  ONE environment variable can
override them for a whole process without touching code — for A/B runs and for harnesses that construct the objects
themselves (the reference's ``Benchmark`` / ``StreamingInference`` build pipelines from a config object):

    DZ_ENGINE="recurrence=3,lanes=6,inflight=7,wait=block,precision=f32,warmup=0,split_strict=0,affinity=0"

Unknown keys are refused; every override that takes effect is logged once to stderr.
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Dict, Optional

KEYS = ("recurrence", "lanes", "inflight", "wait", "precision", "warmup", "split_strict", "affinity", "concurrent_files")
_parsed: Dict[str, Dict[str, str]] = {}
_logged: set = set()


def overrides() -> Dict[str, str]:
    """``DZ_ENGINE`` parsed (cached per value of the variable)."""
    raw = os.environ.get("DZ_ENGINE", "")
    got = _parsed.get(raw)
    if got is None:
        got = {}
        for item in filter(None, (p.strip() for p in raw.split(","))):
            k, sep, v = item.partition("=")
            k = k.strip()
            if not sep or k not in KEYS:
                raise ValueError(f"DZ_ENGINE: {item!r} is not key=value with a key of {KEYS}")
            got[k] = v.strip()
        _parsed[raw] = got
    return got


def setting(key: str, given, default, cast: Optional[Callable] = None):
    """The value of engine parameter ``key``: the ``DZ_ENGINE`` override if there is one (logged once), else the
    constructor argument ``given`` unless it is None, else ``default``."""
    assert key in KEYS, key
    ov = overrides().get(key)
    if ov is not None:
        val = cast(ov) if cast else ov
        if (key, ov) not in _logged:
            _logged.add((key, ov))
            print(f"[diart_amd] DZ_ENGINE overrides {key}={ov!r}" + (f" (constructor argument: {given!r})" if given is not None else ""),
                  file=sys.stderr, flush=True)
        return val
    return default if given is None else given
