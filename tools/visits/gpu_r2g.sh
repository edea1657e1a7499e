#!/bin/bash
# Round-2 GPU visit G: everything (kernel / model / pipeline tests), kbench of conv0, bench grid
#   grid entries: lstm,depth,conv0split,pre
TAG=${1:-r2g}
GRID=${2:-"valu,2,0,1 valu,2,1,1 valu,2,1,0 0,3,1,1"}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
echo "=== tests" >> $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 800 -p no:cacheprovider 2>&1 | tail -30 >> $OUT
echo "=== kbench" >> $OUT
timeout 300 python tools/kbench.py --only ${KB:-sinc_conv0,sinc_conv0_split,wave_stats} 2>&1 | grep -v amdgpu.ids | tail -12 >> $OUT
for cfg in $GRID; do
  IFS=, read l d c0 pre <<< "$cfg"
  sh=1; [ "$l" = "valu" ] && sh=0
  echo "=== bench lstm=$l depth=$d conv0_split=$c0 pre=$pre shared_emb=$sh" >> $OUT
  DZ_CONV0_SPLIT=$c0 DZ_GEMM_PRE=$pre DZ_SHARED_EMB=$sh DZ_LSTM=$l DZ_DEPTH=$d timeout 300 python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-exact-f32 \
      > gpurun_out/bench_${TAG}_${l}_${d}_${c0}_${pre}.json 2>gpurun_out/bench_${TAG}_${l}_${d}_${c0}_${pre}.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_${l}_${d}_${c0}_${pre}.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "host_fed", (d.get("host_fed") or {}).get("value"))
    for k in d["roofline_kernels"][:12]:
        print("   %-40s %7.1f us x%5.2f/step  cpl %5.1f  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["chunks_per_launch"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  grep "timed region" gpurun_out/bench_${TAG}_${l}_${d}_${c0}_${pre}.err | cut -c1-200 >> $OUT
  tail -1 gpurun_out/bench_${TAG}_${l}_${d}_${c0}_${pre}.err | cut -c1-300 >> $OUT
done
cat $OUT | cut -c1-200
