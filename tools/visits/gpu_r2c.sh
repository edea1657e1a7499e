#!/bin/bash
# Round-2 GPU visit C: kernel tests (new pre-split GEMM path), isolated timings, model / pipeline
# parity with the path on, bench with the path on / off.
TAG=${1:-r2c}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
if [ -z "$SKIP_TESTS" ]; then
echo "=== kernel tests" >> $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 500 -p no:cacheprovider 2>&1 | tail -30 >> $OUT
echo "=== model / pipeline tests" >> $OUT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_pipeline.py tests/test_gpu_der.py -m gpu -q --timeout 800 -p no:cacheprovider 2>&1 | tail -30 >> $OUT
echo "=== kbench" >> $OUT
timeout 300 python tools/kbench.py --only lstm_proj,seg_mlp0,tdnn2,tdnn3,tdnn4,tdnn5 2>&1 | grep -v amdgpu.ids | tail -40 >> $OUT
cp gpurun_out/kbench.json gpurun_out/kbench_$TAG.json 2>/dev/null
fi
GRID=${2:-"valu,2,1,0 valu,2,1,1 valu,1,2,1 0,1,3,1 0,1,4,1"}
for cfg in $GRID; do
  IFS=, read l s d pre <<< "$cfg"
  echo "=== bench lstm=$l seg_split=$s depth=$d pre=$pre" >> $OUT
  DZ_GEMM_PRE=$pre DZ_LSTM=$l DZ_SEG_SPLIT=$s DZ_DEPTH=$d timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 \
      > gpurun_out/bench_${TAG}_${l}_${s}_${d}_${pre}.json 2>gpurun_out/bench_${TAG}_${l}_${s}_${d}_${pre}.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_${l}_${s}_${d}_${pre}.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "host_fed", (d.get("host_fed") or {}).get("value"))
    for k in d["roofline_kernels"][:9]:
        print("   %-40s %7.1f us x%5.2f/step  cpl %5.1f  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["chunks_per_launch"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  tail -2 gpurun_out/bench_${TAG}_${l}_${s}_${d}_${pre}.err >> $OUT
done
cat $OUT | cut -c1-200
