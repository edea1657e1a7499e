#!/bin/bash
# One GPU-box visit for a round of kernel work: kernel + pipeline parity tests, isolated kernel
# timings, then bench.py under a grid of StreamBatch sub-batch splits.
#   usage: tools/gpu_sweep.sh <tag> ["seg,emb seg,emb ..."]
TAG=${1:-s1}
GRID=${2:-"1,1 2,1 2,2 4,2"}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/sweep_$TAG.log
: > $OUT
echo "=== tests" >> $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_models.py -m gpu -q --timeout 500 \
    -p no:cacheprovider 2>&1 | tail -25 >> $OUT
echo "=== kbench" >> $OUT
timeout 300 python tools/kbench.py 2>&1 | tail -24 >> $OUT
cp gpurun_out/kbench.json gpurun_out/kbench_$TAG.json 2>/dev/null
for g in $GRID; do
  s=${g%,*}; e=${g#*,}
  echo "=== bench seg_split=$s emb_split=$e" >> $OUT
  DZ_SEG_SPLIT=$s DZ_EMB_SPLIT=$e timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline \
      --kernel-table gpurun_out/kernels_${TAG}_${s}_${e}.json 2>gpurun_out/bench_${TAG}_${s}_${e}.err | tee -a $OUT | cut -c1-200
  tail -2 gpurun_out/bench_${TAG}_${s}_${e}.err >> $OUT
done
grep -E "^===|passed|failed|error|us  |\"value\"" $OUT | cut -c1-260
