"""CPU oracle for the diart per-chunk hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the thing timed as "CPU baseline".
The product package (``diart_amd``) never imports this package and has no CPU
fallback: it raises if the HIP library is missing.

Parity status
-------------
* ``functional_ref`` / ``clustering_ref`` restate ``/root/reference/src/diart/
  functional.py``, ``mapping.py`` and ``blocks/clustering.py``.  They are PINNED:
  ``tests/golden/make_golden.py`` runs the reference's own files (loaded by path
  with the tiny ``pyannote.core`` stand-in in ``oracle/pyannote_stub.py``) and
  commits their outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
  checks the restatement against those fixtures.
* ``models_ref`` restates the third-party networks the reference calls at
  ``src/diart/models.py:133`` and ``:262`` (``pyannote.audio`` ``PyanNet`` and
  ``XVectorSincNet``, pinned by the reference only as ``pyannote.audio>=2.1.1``,
  ``setup.cfg:35``; sinc filters from ``asteroid-filterbanks`` ``ParamSincFB``).
  Neither library nor its gated checkpoints are present in the build container
  or in ``/root/reference`` and the reference has no unit tests or golden
  tensors at this boundary, so the *network arithmetic* is **parity unpinned**:
  it follows the published architecture, is shape- and
  parameter-count-checked (1,472,749 / 4,346,366 parameters, 293 / 279 frames)
  and uses the same state-dict key names as the pyannote checkpoints so real
  weights can be dropped in and re-checked later.
"""
