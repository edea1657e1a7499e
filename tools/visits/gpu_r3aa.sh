#!/bin/bash
# exact-f32 path after the tap-minor k order in convgemm: parity, step time, traffic
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_r2.py tests/test_gpu_ecapa.py -m gpu -q -x --timeout 500 -p no:cacheprovider 2>&1 | tail -4
timeout -s KILL 300 python bench.py --gpus 1 --steps 100 --warmup 10 --precision f32 --pmc all --no-cpu-baseline --no-exact-f32 --no-host-pass > gpurun_out/bench_f32tm.json 2> gpurun_out/bench_f32tm.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_f32tm.json"))
tot = sum((k.get("traffic") or 0) * k["launches_per_step"] for k in d["roofline_kernels"])
print("f32 ms", d["ms_per_step"], "value", d["value"], "traffic GB/step %.3f" % (tot / 1e9))
for k in d["roofline_kernels"][:6]:
    print("  %-44s %7.1f us x%5.2f frac %.3f traffic %s alg %s" % (k["kernel"][:44], k["avg_launch_us"], k["launches_per_step"], k["frac"], k.get("traffic") and round(k["traffic"] / 1e6, 1), k.get("alg_bytes_per_launch") and round(k["alg_bytes_per_launch"] / 1e6, 1)))
PY
