"""The one JSON line `bench.py` prints on stdout, kept small enough for any line-capturing driver.

`bench.py` measures much more than the contract's fields (a roofline entry per device kernel, the same
tables for the exact-f32 pass, the host rehearsal, prose notes).  All of that goes to a side file
(`details_file`, default gpurun_out/bench_details.json) and to stderr; the stdout line is assembled here
from the full record and is guaranteed to stay below ``MAX_LINE`` characters: optional keys are dropped,
least important first, until it fits (tests/test_benchline.py runs this on a canned full record).

What is timed is what the reference's Chronometer brackets (/root/reference/src/diart/utils.py:13-43,
/root/reference/src/diart/inference.py:130-135): the pipeline call on a batch of windows.
"""
from __future__ import annotations

import json
from pathlib import Path

MAX_LINE = 4000          # characters; the verdict's bound is 4 KB, the driver's capture 16 018

_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_occupied_cus", "cus_occupied", "avg_launch_us",
              "launches_per_step", "alg_gflop_per_launch", "alg_bytes_per_launch", "traffic", "traffic_over_alg_bytes",
              "share_of_kernel_time")
# dropped in this order when the line is too long (it never is with the fields below; belt and braces)
_OPTIONAL = ("roofline_exact_f32", "host_rehearsal", "host", "roofline_mfma", "step_period_ms", "power")


def _roof(entry, source=None):
    """The contract's roofline keys of one per-kernel entry (None stays None)."""
    if not entry:
        return None
    r = {k: entry.get(k) for k in _ROOF_KEYS if k in entry}
    src = entry.get("traffic_source") or source or ""
    if entry.get("traffic") is not None:
        r["traffic_source"] = "live-pmc" if src.startswith("live") else ("committed-pmc" if src else None)
    return r


def _minmax(xs):
    xs = [x for x in (xs or []) if x is not None]
    return {"min": min(xs), "max": max(xs)} if xs else None


def compact(full, details_file=None):
    """full record (bench.py's `out`) -> the dict printed on stdout."""
    cfg = full.get("config") or {}
    exact = full.get("exact_f32") or {}
    host_fed = full.get("host_fed") or {}
    mfma = full.get("mfma_util_step") or {}
    hbm = full.get("hbm_gbps_step") or {}
    roof = full.get("roofline") or {}
    host = full.get("host") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {
        "workload": str(cfg.get("workload", ""))[:240],
        "streams_per_gpu": cfg.get("streams_per_gpu"), "chunks_per_step": cfg.get("chunks_per_step"),
        "parallelism": cfg.get("parallelism"), "dist_backend": cfg.get("dist_backend"),
        "rccl_ranks": cfg.get("rccl_ranks"),
        "weights_abs_sum_per_rank": _minmax(cfg.get("weights_abs_sum_per_rank")),
        "host_threads_per_rank": cfg.get("host_threads_per_rank"),
        "steps_in_flight": cfg.get("steps_in_flight"), "lanes": cfg.get("lanes"), "recurrence": cfg.get("recurrence"),
        "settle_steps": cfg.get("settle_steps"), "engine": cfg.get("engine"),
        "cpu_affinity": cfg.get("cpu_affinity"),
    }
    for k in ("latency_ms", "chunk_ms", "file_seconds", "wall_s"):           # configs 1 / 5
        if k in full:
            out[k] = full[k]
    out["value_exact_f32"] = exact.get("value")
    out["ms_per_step_exact_f32"] = exact.get("ms_per_step")
    out["value_host_fed"] = host_fed.get("value")
    r = _roof(roof)
    if r is not None:
        r["whole_path_tflops"] = roof.get("whole_path_tflops")
        # serialised: durations from the one-lane / one-stream pass (alone-times, = rocprofv3 --kernel-trace --stats of
        # `bench.py --serial-only`); False: brackets taken while `steps_overlapping` steps shared the chip
        r["serialised"] = bool(roof.get("serialised"))
        if roof.get("serialised"):
            r["serialised_ms_per_step"] = roof.get("serialised_ms_per_step")
        elif cfg.get("lanes"):
            r["steps_overlapping"] = cfg.get("lanes")
    out["roofline"] = r
    out["whole_path_frac"] = full.get("whole_path_frac")
    out["whole_path_frac_exact_f32"] = full.get("whole_path_frac_exact_f32")
    out["roofline_mfma"] = _roof(full.get("roofline_mfma"), roof.get("traffic_source"))
    out["roofline_exact_f32"] = _roof(exact.get("roofline"))
    out["mfma_busy_frac_step"] = mfma.get("busy_frac_pmc")
    out["mfma_issued_frac_of_peak_step"] = mfma.get("frac_of_peak")
    out["hbm_gbps_step"] = hbm.get("gbps")
    out["hbm_frac_step"] = hbm.get("frac_of_peak")
    pw = full.get("power")
    if isinstance(pw, dict):        # what the card drew while the pipeline ran (hwmon): the default precision sits at the power budget
        out["power"] = {k: pw.get(k) for k in ("package_w", "sclk_mhz", "idle_w", "joules_per_step")}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                               "kind": cb.get("kind"), "dedup_value": cb.get("dedup_value"),
                               "sample": str(cb.get("sample", ""))[:200]}
    else:
        out["cpu_baseline"] = None
    out["step_period_ms"] = host.get("step_period_ms_in_timed_region")
    if host:
        out["host"] = {"cpu_ms_per_step": host.get("cpu_ms_per_step"), "launch_ms_per_step": host.get("launch_ms_per_step"),
                       "threads": host.get("threads"), "usable_cores": host.get("usable_cores")}
    hr = full.get("host_rehearsal")
    if isinstance(hr, dict):
        out["host_rehearsal"] = {"ranks_emulated": hr.get("ranks_emulated"), "cores_per_rank": hr.get("cores_per_rank"),
                                 "pinned_value": (hr.get("pinned") or {}).get("value"),
                                 "unpinned_value": (hr.get("unpinned") or {}).get("value"),
                                 "pinned_over_unpinned": hr.get("pinned_over_unpinned")}
    out["details_file"] = details_file
    for k in _OPTIONAL:
        if len(json.dumps(out)) <= MAX_LINE:
            break
        out.pop(k, None)
    return out


def line(full, details_file=None):
    """The stdout line itself (one JSON object, no newline inside, <= MAX_LINE characters)."""
    d = compact(full, details_file)
    s = json.dumps(d)
    if len(s) > MAX_LINE:            # cannot happen with the bounded fields above; never lose the line to it
        core = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "value_exact_f32", "details_file")
        d = {k: (v if not isinstance(v, str) else v[:200]) for k, v in d.items() if k in core}
        s = json.dumps(d)
    return s


def write_details(full, path):
    """Full record -> side file; returns the path written (relative form kept) or None."""
    try:
        p = Path(path)
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(json.dumps(full, indent=1))
        return str(path)
    except OSError:
        return None
