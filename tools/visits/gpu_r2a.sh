#!/bin/bash
# Round-2 GPU visit A: full GPU test suite, isolated kernel timings (VALU vs matrix-core LSTM),
# bench.py under a grid of (seg sub-batches, steps in flight).
#   usage: tools/gpu_r2a.sh <tag> ["split,depth split,depth ..."]
TAG=${1:-r2a}
GRID=${2:-"2,1 2,2 1,2 1,3 1,4"}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
echo "=== tests" >> $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -30 >> $OUT
echo "=== kbench" >> $OUT
timeout 300 python tools/kbench.py 2>&1 | tail -30 >> $OUT
cp gpurun_out/kbench.json gpurun_out/kbench_$TAG.json 2>/dev/null
for g in $GRID; do
  s=${g%,*}; d=${g#*,}
  echo "=== bench seg_split=$s depth=$d" >> $OUT
  DZ_SEG_SPLIT=$s DZ_DEPTH=$d timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 \
      --kernel-table gpurun_out/kernels_${TAG}_${s}_${d}.json > gpurun_out/bench_${TAG}_${s}_${d}.json 2>gpurun_out/bench_${TAG}_${s}_${d}.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_${s}_${d}.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "host_fed", (d.get("host_fed") or {}).get("value"))
    for k in d["roofline_kernels"][:8]:
        print("   %-40s %7.1f us x%5.2f/step  cpl %5.1f  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["chunks_per_launch"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  tail -3 gpurun_out/bench_${TAG}_${s}_${d}.err >> $OUT
done
cat $OUT | cut -c1-220
