"""The host half of the path (csrc/cluster.cpp, tail.cpp, hostpool.cpp, filebatch.cpp: fp64 OnlineSpeakerClustering,
DelayedAggregation + Binarize and the worker pool that runs them for N streams — the reference's
blocks/clustering.py, mapping.py, blocks/aggregation.py, blocks/utils.py) under AddressSanitizer + UBSan and under
ThreadSanitizer.  GPU sanitizers are not available on this pool, and these four files have no HIP in them: they are
compiled here with g++ together with tests/native/host_sanitize.cpp (random streams, NaN embeddings, resets, several
thread counts, two callers sharing the process-wide pool) and must run without a report."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = [ROOT / "tests" / "native" / "host_sanitize.cpp"] + [ROOT / "diart_amd" / "csrc" / f
                                                            for f in ("cluster.cpp", "tail.cpp", "hostpool.cpp", "filebatch.cpp")]


@pytest.mark.parametrize("flags,env", [
    (["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0"}),
    (["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"}),
], ids=["asan+ubsan", "tsan"])
def test_host_half_is_clean_under_sanitizers(tmp_path, flags, env):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "host_sanitize"
    build = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", *flags, *map(str, SRC), "-o", str(exe), "-lpthread"],
                           capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip("sanitizer runtime not installed: " + build.stderr.strip().splitlines()[-1])
    assert build.returncode == 0, build.stderr[-2000:]
    import os
    run = subprocess.run([str(exe), "40"], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    if "unexpected memory mapping" in run.stderr or "ReExec" in run.stderr:
        pytest.skip("the sanitizer runtime cannot map its shadow memory on this kernel (ASLR setting): " + run.stderr.strip().splitlines()[0])
    assert run.returncode == 0 and "host_sanitize ok" in run.stdout, (run.stdout[-500:] + run.stderr[-3000:])
    assert "ERROR: " not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
