"""The multi-rank path on real hardware.

* RCCL itself: on a one-GPU box two ranks cannot share the device (RCCL refuses duplicate GPUs), so
  the RCCL broadcast / barrier / all_reduce(MAX) of bench.py's weight and timing paths run on a
  group of ONE rank on cuda:0 — the collectives are real RCCL kernels either way.
* `python bench.py --gpus 2` as ONE process starts its own 2 ranks: over RCCL when the node has
  >= 2 GPUs, else as the documented single-GPU rehearsal (gloo + DZ_FORCE_DEVICE=0)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def test_rccl_broadcast_and_timing_bracket_on_cuda0(tmp_path):
    from _dist_scripts import ONE_RANK_GROUP
    script = tmp_path / "one.py"
    script.write_text(ONE_RANK_GROUP)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION")
    r = subprocess.run([sys.executable, str(script), str(ROOT), "nccl"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "group of one ok: nccl" in r.stdout, r.stdout + r.stderr


def test_bench_started_as_one_process_runs_two_ranks():
    two_gpus = torch.cuda.device_count() >= 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if not two_gpus:
        env.update(DZ_FORCE_DEVICE="0", DZ_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--streams", "64", "--no-cpu-baseline", "--no-exact-f32", "--no-host-pass"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["chunks_per_step"] == 128
    assert out["config"]["dist_backend"] == ("nccl" if two_gpus else "gloo")
    ws = out["config"]["weights_abs_sum_per_rank"]                    # compact line: {min, max} over the ranks
    assert ws["min"] == ws["max"] > 0                                 # identical weights on both ranks
    assert out["config"]["rccl_ranks"] == (2 if two_gpus else 0) and len(lines[0]) < 4096
    assert "process group up" in r.stderr and out["value"] > 0


def test_bench_refuses_more_ranks_than_gpus():
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DZ_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
