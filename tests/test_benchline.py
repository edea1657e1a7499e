"""The stdout line of bench.py: one JSON object, <= 4000 characters, whatever the full record holds.
(Round 4's line grew to 21 KB and the driver could not parse it; the full record used here is that very line.)"""
import copy
import json
from pathlib import Path

import pytest

from diart_amd import benchline
from diart_amd import hostinfo

ROOT = Path(__file__).resolve().parent.parent
FULL = json.loads((ROOT / "profiles" / "r04_g_bench_driver_form.json").read_text())


def test_canned_full_record_gives_a_short_parseable_line():
    assert len(json.dumps(FULL)) > 16018                      # the record that broke the driver's capture
    s = benchline.line(FULL, "gpurun_out/bench_details.json")
    assert len(s) < 4096 and "\n" not in s
    d = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == FULL["value"] and d["ms_per_step"] == FULL["ms_per_step"]
    assert d["value_exact_f32"] == FULL["exact_f32"]["value"]
    assert d["ms_per_step_exact_f32"] == FULL["exact_f32"]["ms_per_step"]
    r = d["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step",
              "traffic_over_alg_bytes"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-2)
    assert r["traffic_source"] == "live-pmc"
    assert d["roofline_mfma"]["kernel"].startswith("gemm_pre_kernel")
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 16
    assert d["step_period_ms"] == FULL["host"]["step_period_ms_in_timed_region"]
    assert d["config"]["weights_abs_sum_per_rank"]["min"] == d["config"]["weights_abs_sum_per_rank"]["max"]
    assert d["details_file"] == "gpurun_out/bench_details.json"


def test_eight_rank_record_and_pathological_strings_still_fit():
    full = copy.deepcopy(FULL)
    full["n_gpus"] = 8
    full["config"]["weights_abs_sum_per_rank"] = [519213.39691358176] * 8
    full["config"]["dist_backend"], full["config"]["rccl_ranks"] = "nccl", 8
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["roofline_kernels"] = full["roofline_kernels"] * 10
    d = json.loads(benchline.line(full, None))
    assert d["config"]["rccl_ranks"] == 8 and d["config"]["weights_abs_sum_per_rank"] == {"min": 519213.39691358176, "max": 519213.39691358176}
    assert len(json.dumps(d)) < 4096


def test_power_entry_reaches_the_line():
    """bench.py's hwmon reading of the pass behind the timed region (package watts, shader clock): four numbers in the line,
    the prose stays in the details file."""
    full = copy.deepcopy(FULL)
    full["power"] = {"package_w": 1365, "sclk_mhz": 2032, "idle_w": 242, "ms_per_step": 0.846, "joules_per_step": 1.155,
                     "seconds": 1.5, "steps": 1750, "samples": 44, "source": "x" * 500}
    d = json.loads(benchline.line(full, None))
    assert d["power"] == {"package_w": 1365, "sclk_mhz": 2032, "idle_w": 242, "joules_per_step": 1.155}
    assert len(json.dumps(d)) < 4096


def test_missing_parts_do_not_break_the_line():
    d = json.loads(benchline.line({"metric": "m", "value": 1.0}, None))
    assert d["roofline"] is None and d["cpu_baseline"] is None and d["value"] == 1.0


def test_details_file_round_trips(tmp_path):
    p = tmp_path / "sub" / "d.json"
    assert benchline.write_details(FULL, p) == str(p)
    assert json.loads(p.read_text()) == FULL


def test_rank_cpu_share_partitions_near_cores():
    # two sockets: GPUs 0-3 near cpus 0-15, GPUs 4-7 near 16-31; everything allowed
    local = lambda g: list(range(0, 16)) if g < 4 else list(range(16, 32))
    shares = [hostinfo.rank_cpu_share(r, 8, range(32), local_of=local) for r in range(8)]
    assert shares[0] == [0, 1, 2, 3] and shares[3] == [12, 13, 14, 15] and shares[4] == [16, 17, 18, 19]
    assert sorted(c for s in shares for c in s) == list(range(32))
    # unknown topology: an even split of what is allowed
    shares = [hostinfo.rank_cpu_share(r, 8, range(16), local_of=lambda g: []) for r in range(8)]
    assert shares == [[2 * r, 2 * r + 1] for r in range(8)]
    # fewer near cores than ranks: leave the affinity alone
    assert hostinfo.rank_cpu_share(0, 8, range(4), local_of=lambda g: []) == []
    # rehearsal: every rank on GPU 0
    shares = [hostinfo.rank_cpu_share(r, 2, range(8), local_of=lambda g: [0, 1, 2, 3], device_of=lambda r: 0) for r in range(2)]
    assert shares == [[0, 1], [2, 3]]
    assert hostinfo.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_bind_rank_never_raises_and_is_a_noop_for_one_rank():
    assert hostinfo.bind_rank(0, 1) == {"bound": False, "cpus": None}
