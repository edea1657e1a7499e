#!/bin/bash
# One GPU-box visit: PMC traffic passes, bench line (+cpu baseline), then the same command under
# rocprofv3 --kernel-trace --stats.   usage: tools/gpu_bench.sh <tag> [steps]
TAG=${1:-r02_a}
STEPS=${2:-40}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
REPO=$PWD
bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1
tail -12 gpurun_out/pmc_$TAG.txt
cp gpurun_out/traffic_$TAG.json profiles/traffic.json
cp gpurun_out/mfma_$TAG.json profiles/mfma_util.json
timeout 420 python bench.py --steps $STEPS --warmup 5 --kernel-table gpurun_out/kernels_$TAG.json \
    > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit $?"
cat gpurun_out/bench_$TAG.json
tail -3 gpurun_out/bench_$TAG.err
cd /tmp
DZ_PROF_EVERY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- \
    python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-host-pass --kernel-table $REPO/gpurun_out/kernels_${TAG}_same_run.json > $REPO/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit $?"
cd $REPO
find gpurun_out/prof_$TAG -name '*kernel_trace*' -size +2M -delete
ls -R gpurun_out/prof_$TAG | head
