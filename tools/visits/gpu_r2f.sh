#!/bin/bash
# Round-2 GPU visit F: shared embedding stream + pooling lag; grid entries hwq,lstm,split,depth,shared
TAG=${1:-r2f}
GRID=${2:-"8,valu,1,2,1 8,valu,1,2,0 8,valu,1,3,1 8,0,1,3,1 8,0,1,4,1 8,0,1,6,1 8,0,2,3,1"}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
if [ -n "$RUN_TESTS" ]; then
echo "=== pipeline tests" >> $OUT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_der.py -m gpu -q --timeout 800 -p no:cacheprovider 2>&1 | tail -30 >> $OUT
fi
for cfg in $GRID; do
  IFS=, read q l s d sh <<< "$cfg"
  echo "=== bench hwq=$q lstm=$l seg_split=$s depth=$d shared_emb=$sh" >> $OUT
  GPU_MAX_HW_QUEUES=$q DZ_SHARED_EMB=$sh DZ_LSTM=$l DZ_SEG_SPLIT=$s DZ_DEPTH=$d timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-exact-f32 \
      > gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${sh}.json 2>gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${sh}.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${sh}.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "host_fed", (d.get("host_fed") or {}).get("value"))
    for k in d["roofline_kernels"][:5]:
        print("   %-40s %7.1f us x%5.2f/step  cpl %5.1f  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["chunks_per_launch"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  grep "timed region" gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${sh}.err | cut -c1-200 >> $OUT
  tail -1 gpurun_out/bench_${TAG}_${q}_${l}_${s}_${d}_${sh}.err | cut -c1-200 >> $OUT
done
cat $OUT | cut -c1-200
