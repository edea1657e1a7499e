"""Build libdiart_amd.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

``python -m diart_amd.build`` or ``diart_amd.build.build()``.  The shared object is
written next to this file so it travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libdiart_amd.so"
LIB_EXPERIMENTS = HERE / "libdiart_amd_exp.so"
SOURCES = ["api.hip", "ecapa_api.hip", "k_front.hip", "k_convgemm.hip", "k_gemm_f32.hip", "k_gemm_split.hip", "k_gemm_pre.hip",
           "k_mlp_head.hip", "k_conv_pool.hip", "k_lstm.hip", "k_lstm_mfma.hip", "k_pool.hip",
           "k_ecapa.hip", "ring.hip", "cluster.cpp", "tail.cpp", "hostpool.cpp", "filebatch.cpp"]
# -DDZ_EXPERIMENTS only (csrc/dz_common.h "build flavours"): the never-default GEMM generations
EXPERIMENT_SOURCES = ["experiments/k_gemm_g2.hip", "experiments/k_gemm_g3.hip"]
ARCH = "gfx950"
# per-source flags.  k_lstm_mfma.hip: the cell update beside the recurrence's MFMAs stays plain f32 instructions (hipcc's
# SLP vectoriser would pack adjacent adds / multiplies into v_pk_*_f32, which cost more beside MFMAs than the two they replace)
EXTRA_FLAGS = {"k_lstm_mfma.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> Path:
    """experiments=False: the shipped library (one configuration per layer, no kernel-selection switches).
    experiments=True: libdiart_amd_exp.so with -DDZ_EXPERIMENTS (measurement builds; DZ_EXPERIMENTS=1 loads it)."""
    hipcc = _hipcc()
    objdir = HERE / "build" / ("exp" if experiments else "ship")
    objdir.mkdir(parents=True, exist_ok=True)
    headers = [CSRC / "dz_common.h", CSRC / "hostpool.h", HERE.parent / "include" / "diart_amd.h",
               HERE.parent / "include" / "diart_amd_experiments.h"]
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", f"-I{CSRC}"]
    sources, lib = list(SOURCES), LIB
    if experiments:
        flags.append("-DDZ_EXPERIMENTS")
        sources, lib = sources + EXPERIMENT_SOURCES, LIB_EXPERIMENTS

    def compile_one(src: str) -> Path:
        s = CSRC / src
        o = objdir / (s.stem + ".o")
        if force or _stale(o, [s, *headers]):
            cmd = [hipcc, *flags, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", str(s), "-o", str(o)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        return o

    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, sources))
    if force or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(lib), "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
