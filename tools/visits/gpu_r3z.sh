#!/bin/bash
# tap-minor k order in gemm_pre + row-tile group size of the pooled tdnn5: parity tests, traffic per launch (live PMC), step time
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_pipeline.py -m gpu -q -x --timeout 500 -p no:cacheprovider 2>&1 | tail -5
for k in 0 2 1 8; do
  DZ_POOL_AG=$k timeout -s KILL 300 python bench.py --gpus 1 --steps 200 --warmup 10 --pmc all --no-cpu-baseline --no-exact-f32 --no-host-pass > gpurun_out/bench_ag_$k.json 2> gpurun_out/bench_ag_$k.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_ag_$k.json"))
tot = sum((k.get("traffic") or 0) * k["launches_per_step"] for k in d["roofline_kernels"])
print("POOL_AG=$k ms", d["ms_per_step"], "traffic GB/step %.3f" % (tot / 1e9), " ".join("%s=%.0fMB/%.0fus" % (k["kernel"][:20], (k.get("traffic") or 0) / 1e6, k["avg_launch_us"]) for k in d["roofline_kernels"] if "gemm_pre" in k["kernel"]))
PY
done
