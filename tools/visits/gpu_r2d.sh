#!/bin/bash
# Round-2 GPU visit D: does the runtime's hardware-queue limit serialise the lanes?
#   grid entries: hwq,lstm,split,depth,pre
TAG=${1:-r2d}
GRID=${2:-"4,valu,2,1,1 8,valu,2,1,1 8,valu,2,2,1 8,valu,1,2,1 8,valu,1,3,1 16,valu,2,2,1 16,valu,1,3,1 8,0,1,3,1 16,0,1,4,1 16,0,2,3,1"}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
for cfg in $GRID; do
  IFS=, read q l s d pre <<< "$cfg"
  echo "=== bench hwq=$q lstm=$l seg_split=$s depth=$d pre=$pre" >> $OUT
  GPU_MAX_HW_QUEUES=$q DZ_GEMM_PRE=$pre DZ_LSTM=$l DZ_SEG_SPLIT=$s DZ_DEPTH=$d timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-exact-f32 \
      > gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${pre}.json 2>gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${pre}.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${pre}.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "host_fed", (d.get("host_fed") or {}).get("value"))
    for k in d["roofline_kernels"][:5]:
        print("   %-40s %7.1f us x%5.2f/step  cpl %5.1f  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["chunks_per_launch"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  grep "timed region" gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${pre}.err | cut -c1-200 >> $OUT
done
cat $OUT | cut -c1-200
