#!/bin/bash
# Round-2 GPU visit B: instruction-rate micro-benchmarks, LSTM variants (tests + isolated timing),
# bench.py with the VALU recurrence under (split, depth) and the matrix-core variants.
TAG=${1:-r2b}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
echo "=== ubench" >> $OUT
timeout 120 tools/ubench/ubench >> $OUT 2>&1
echo "=== lstm tests" >> $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k lstm --timeout 500 -p no:cacheprovider 2>&1 | tail -15 >> $OUT
echo "=== kbench lstm" >> $OUT
timeout 300 python tools/kbench.py --only lstm,lstm_mfma0,lstm_mfma1,lstm_mfma2 2>&1 | tail -12 >> $OUT
for cfg in "valu,2,1" "valu,2,2" "valu,1,2" "1,1,2" "2,1,2" "1,1,4"; do
  IFS=, read l s d <<< "$cfg"
  echo "=== bench lstm=$l seg_split=$s depth=$d" >> $OUT
  DZ_LSTM=$l DZ_SEG_SPLIT=$s DZ_DEPTH=$d timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 \
      > gpurun_out/bench_${TAG}_${l}_${s}_${d}.json 2>gpurun_out/bench_${TAG}_${l}_${s}_${d}.err
  python - <<PY >> $OUT
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_${l}_${s}_${d}.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "host_fed", (d.get("host_fed") or {}).get("value"))
    for k in d["roofline_kernels"][:4]:
        print("   %-40s %7.1f us x%5.2f/step  cpl %5.1f  %8.2f %s frac %.3f share %.3f" % (k["kernel"][:40], k["avg_launch_us"], k["launches_per_step"], k["chunks_per_launch"], k["achieved"], k["unit"], k["frac"], k["share_of_kernel_time"]))
except Exception as e:
    print("bench failed:", e)
PY
  tail -2 gpurun_out/bench_${TAG}_${l}_${s}_${d}.err >> $OUT
done
cat $OUT | cut -c1-200
