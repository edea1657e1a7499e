"""Shader clock and package power of the GPU under test, read from the amdgpu hwmon files by a side thread.

Measurement helper of ``bench.py`` (its ``power`` entry) and ``tools/kbench.py --power``; nothing on the product path
imports it.  A one-GPU box still lists every card of its node in sysfs: the card under test is the one whose power
moves while the sampler runs."""
from __future__ import annotations

import glob
import math
import os
import statistics
import threading
import time
from typing import List, Optional, Tuple


def pci_address(device_index: int = 0) -> Optional[str]:
    """PCI address ("0000:05:00.0") of HIP device ``device_index`` of this process (hipDeviceGetPCIBusId), or None."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
            return None
        return buf.value.decode().strip().lower() or None
    except Exception:      # noqa: BLE001 — a measurement helper never takes the bench down
        return None


class PowerSampler:
    """``freq1_input`` (Hz) and ``power1_average`` / ``power1_input`` (uW) of every card, every ``period`` seconds.
    ``device_index``: the HIP device under test; its card is found through its PCI address (the sysfs `device` link of a
    card resolves to it).  Without a match — or without a device index — the card whose power moved most is taken, which on
    a shared node can be another tenant's."""

    def __init__(self, period: float = 0.02, device_index: Optional[int] = None):
        self.freq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        self.own: Optional[int] = None
        addr = pci_address(device_index) if device_index is not None else None
        if addr:
            for i, f in enumerate(self.freq):
                dev = os.path.realpath(f.split("/hwmon/")[0])
                if os.path.basename(dev).lower() == addr:
                    self.own = i
                    break
        self.power = []
        for f in self.freq:                      # the package power file's name differs between driver versions
            cand = [f.replace("freq1_input", n) for n in ("power1_average", "power1_input")]
            self.power.append(next((c for c in cand if os.path.exists(c)), cand[0]))
        self.period = period
        self.rows: List[Tuple[float, List[float], List[float]]] = []
        self._stop = False
        self._thread = threading.Thread(target=self._run, daemon=True)
        if self.freq:
            self._thread.start()

    @property
    def available(self) -> bool:
        return bool(self.freq)

    @staticmethod
    def _read(path: str) -> float:
        try:
            with open(path) as fh:
                return float(fh.read())
        except (OSError, ValueError):
            return float("nan")

    def _run(self) -> None:
        nan = float("nan")
        while not self._stop:
            if self.own is not None:             # only the card under test: two reads per sample
                fr, pw = [nan] * len(self.freq), [nan] * len(self.freq)
                fr[self.own], pw[self.own] = self._read(self.freq[self.own]) / 1e6, self._read(self.power[self.own]) / 1e6
            else:
                fr, pw = [self._read(f) / 1e6 for f in self.freq], [self._read(p) / 1e6 for p in self.power]
            self.rows.append((time.time(), fr, pw))
            time.sleep(self.period)

    def stop(self) -> None:
        self._stop = True
        if self._thread.is_alive():
            self._thread.join()

    def card(self) -> int:
        """Index of the card under test: by PCI address when known, else the one whose power moved most so far."""
        if self.own is not None:
            return self.own
        span = []
        for c in range(len(self.freq)):
            v = [r[2][c] for r in self.rows if not math.isnan(r[2][c])]
            span.append(max(v) - min(v) if v else 0.0)
        return max(range(len(span)), key=span.__getitem__) if span else 0

    def window(self, t0: float, t1: float, card: Optional[int] = None) -> Tuple[Optional[float], Optional[float], int]:
        """(median MHz, median W, samples) of ``card`` between the wall-clock times t0 and t1."""
        card = self.card() if card is None else card
        w = [r for r in self.rows if t0 <= r[0] <= t1 and not math.isnan(r[2][card])]
        if not w:
            return None, None, 0
        return statistics.median(r[1][card] for r in w), statistics.median(r[2][card] for r in w), len(w)
