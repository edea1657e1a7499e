#!/bin/bash
# what bounds the step: each network alone on the chip, lanes / in-flight variations (timing only)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
TAG=r4d
i=0
for cfg in DZ_ABLATE=noemb DZ_ABLATE=noemb,DZ_DEPTH=3,DZ_INFLIGHT=4 DZ_ABLATE=noemb,DZ_DEPTH=4,DZ_INFLIGHT=5 DZ_ABLATE=noemb,DZ_DEPTH=1 DZ_ABLATE=noemb,DZ_GEMM_GEN=2; do
  i=$((i+1))
  env $(echo $cfg | tr ',' ' ') timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact-f32 --pmc off --no-host-pass \
      > gpurun_out/bench_${TAG}_$i.json 2>gpurun_out/bench_${TAG}_$i.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_$i.json"))
    print("$cfg value", d["value"], "ms/step", d["ms_per_step"], " | ".join("%s %.0f" % (k["kernel"][:18], k["avg_launch_us"]) for k in d["roofline_kernels"][:10]))
except Exception as e:
    print("$cfg bench failed:", e)
PY
  grep "timed region" gpurun_out/bench_${TAG}_$i.err | cut -c1-220
done
