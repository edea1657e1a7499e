"""Rectangular linear sum assignment through libdiart_amd's ``dz_lsap`` (the C++ port of the
algorithm scipy uses, checked against scipy in tests/test_clustering.py); host-only, no GPU."""
from __future__ import annotations

import numpy as np

from . import _lib


def linear_sum_assignment(cost: np.ndarray, maximize: bool = False):
    cost = np.ascontiguousarray(-np.asarray(cost, dtype=np.float64) if maximize
                                else np.asarray(cost, dtype=np.float64))
    nr, nc = cost.shape
    col4row = np.empty(nr, dtype=np.int32)
    _lib.check(_lib.load().dz_lsap(cost.ctypes.data, nr, nc, col4row.ctypes.data), "dz_lsap")
    rows = np.nonzero(col4row >= 0)[0]
    return rows, col4row[rows].astype(np.int64)
