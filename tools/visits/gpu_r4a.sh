#!/bin/bash
# Round-4 visit A: generation-2 GEMM — parity tests, g2bench (alone / beside recurrence workgroups), pipeline A/B.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/visit_r4a.log
: > $OUT
echo "=== test_gemm_pre (all generations)" >> $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm_pre" -p no:cacheprovider 2>&1 | tail -15 >> $OUT
echo "=== g2bench" >> $OUT
timeout 400 python tools/g2bench.py --out gpurun_out/g2bench_r4a.json 2>&1 | grep -v amdgpu.ids | tail -20 >> $OUT
echo "=== network tests with DZ_GEMM_GEN=2" >> $OUT
DZ_GEMM_GEN=2 timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_der.py tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 >> $OUT
i=0
for cfg in DZ_GEMM_GEN=1 DZ_GEMM_GEN=2 DZ_GEMM_GEN=2,DZ_G2_MT=3 DZ_GEMM_GEN=1 DZ_GEMM_GEN=2 DZ_GEMM_GEN=2,DZ_G2_MT=4; do
  i=$((i+1))
  echo "=== bench $cfg" >> $OUT
  env $(echo $cfg | tr ',' ' ') timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact-f32 --pmc off --no-host-pass \
      > gpurun_out/bench_r4a_$i.json 2>gpurun_out/bench_r4a_$i.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_r4a_$i.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"])
    for k in d["roofline_kernels"][:16]:
        print("   %-44s %7.1f us x%5.2f/step  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:44], k["avg_launch_us"], k["launches_per_step"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  tail -2 gpurun_out/bench_r4a_$i.err | cut -c1-300 >> $OUT
done
cat $OUT | cut -c1-260
