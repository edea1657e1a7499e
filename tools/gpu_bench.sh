#!/bin/bash
# One GPU-box visit: bench line (+cpu baseline), then the same command under rocprofv3.
# usage: tools/gpu_bench.sh <tag> [steps]
TAG=${1:-r1}
STEPS=${2:-30}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
timeout 420 python bench.py --steps $STEPS --warmup 5 --kernel-table gpurun_out/kernels_$TAG.json \
    > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit $?"
cat gpurun_out/bench_$TAG.json
tail -5 gpurun_out/bench_$TAG.err
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- \
    python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $REPO/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit $?"
cd $REPO
ls -R gpurun_out/prof_$TAG | head -20
# keep only the small summaries
find gpurun_out/prof_$TAG -name '*kernel_trace*' -size +2M -delete
tail -3 gpurun_out/prof_$TAG.log
