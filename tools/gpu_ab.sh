#!/bin/bash
# Round-2 GPU visit K: kernel tests, kbench subset, bench A/B over env toggles.
#   usage: tools/gpu_ab.sh <tag> "<kbench --only list>" "ENV1=a,ENV2=b ENV1=c ..."   (each grid entry: comma-separated env assignments)
TAG=${1:-r2k}
KB=${2:-conv1_pool,conv2_pool}
GRID=${3:-"DZ_CONV_POOL=1 DZ_CONV_POOL=0 DZ_CONV_POOL=1 DZ_CONV_POOL=0"}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
if [ -z "$SKIP_TESTS" ]; then
echo "=== tests" >> $OUT
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_pipeline.py} -m gpu -q --timeout 800 -p no:cacheprovider 2>&1 | tail -30 >> $OUT
fi
if [ "$KB" != "none" ]; then
echo "=== kbench" >> $OUT
timeout 300 python tools/kbench.py --only $KB 2>&1 | grep -v amdgpu.ids | tail -24 >> $OUT
fi
i=0
for cfg in $GRID; do
  i=$((i+1))
  echo "=== bench $cfg" >> $OUT
  env $(echo $cfg | tr ',' ' ') timeout ${BENCH_TIMEOUT:-90} python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-exact-f32 --pmc off ${BENCH_ARGS:---no-host-pass} \
      > gpurun_out/bench_${TAG}_$i.json 2>gpurun_out/bench_${TAG}_$i.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_$i.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "host_fed", (d.get("host_fed") or {}).get("value"))
    for k in d["roofline_kernels"][:${NK:-14}]:
        print("   %-40s %7.1f us x%5.2f/step  cpl %5.1f  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["chunks_per_launch"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  grep "timed region (" gpurun_out/bench_${TAG}_$i.err | cut -c1-200 >> $OUT
  tail -1 gpurun_out/bench_${TAG}_$i.err | cut -c1-300 >> $OUT
done
cat $OUT | cut -c1-200
