#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 300 python bench.py --config 3 --steps 20 --warmup 3 2> gpurun_out/bench_r4o_config3.err | tail -1 > gpurun_out/bench_r4o_config3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r4o_config3.json"))
print("config3 value", d["value"], "ms/step", d["ms_per_step"], "kernel ms/step", d["roofline"].get("kernel_time_ms_per_step"))
for g in d["roofline_kernels"]:
    print("  %-16s %6.3f ms/step  x%5.1f  %7.1f us  %s %s frac %s share %.3f" % (g["kernel"], g["ms_per_step"], g["launches_per_step"], g["avg_launch_us"], g.get("achieved"), g.get("unit", ""), g.get("frac"), g["share_of_kernel_time"]))
PY
tail -3 gpurun_out/bench_r4o_config3.err | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_ecapa.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
