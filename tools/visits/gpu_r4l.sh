#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_long_horizon.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | cut -c1-300 > gpurun_out/long_horizon_r4l.txt
tail -12 gpurun_out/long_horizon_r4l.txt
echo "=== bench driver form"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc all > gpurun_out/bench_r4l_driver.json 2> gpurun_out/bench_r4l_driver.err
echo "exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r4l_driver.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "exact", d["exact_f32"]["value"], "host_fed", d["host_fed"]["value"])
print("mfma_util_step", json.dumps(d["mfma_util_step"])[:400])
print("hbm_gbps_step", json.dumps(d["hbm_gbps_step"])[:300])
print("roofline_mfma", {k: d["roofline_mfma"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "mfma_util_pmc", "traffic_over_alg_bytes") if k in d["roofline_mfma"]})
print("host", d["host"])
print("host_rehearsal", json.dumps(d["host_rehearsal"])[:900])
print("cpu_baseline", json.dumps(d["cpu_baseline"])[:500])
PY
grep "timed region\|host work\|pmc:" gpurun_out/bench_r4l_driver.err | cut -c1-250
