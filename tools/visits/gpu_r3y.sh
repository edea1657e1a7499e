#!/bin/bash
# driver-form bench with the live PMC passes: where does the HBM traffic stand after the kb-major planes?
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc all > gpurun_out/bench_r3y_driver.json 2> gpurun_out/bench_r3y_driver.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r3y_driver.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"])
print("exact_f32", {k: d["exact_f32"].get(k) for k in ("value", "ms_per_step")})
tot = 0
for k in d["roofline_kernels"]:
    t = k.get("traffic")
    if t:
        tot += t * k["launches_per_step"]
    print("  %-40s %7.1f us x%5.2f  frac %.3f  traffic/launch %s alg %s" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["frac"], t and round(t / 1e6, 1), round(k.get("alg_bytes_per_launch", 0) / 1e6, 1) if k.get("alg_bytes_per_launch") else None))
print("sum traffic per step GB", tot / 1e9)
PY
tail -3 gpurun_out/bench_r3y_driver.err | cut -c1-250
