#!/bin/bash
# One GPU-box visit: the bench line (+cpu baseline), the same command under rocprofv3 --kernel-trace
# --stats, then the PMC passes (HBM traffic, matrix-core busy).   usage: tools/gpu_bench.sh <tag> [steps]
# Order matters on this pool: twice a box whose FIRST GPU process was a rocprofv3 --pmc run died with
# "Memory access fault by GPU" and stayed unusable for the following plain runs; with a plain run
# first the same passes went through (tools/gpu_diag.sh).  Every stage is under a hard timeout.
TAG=${1:-r02_a}
STEPS=${2:-200}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
REPO=$PWD
timeout -s KILL 300 python bench.py --steps $STEPS --warmup 10 --kernel-table gpurun_out/kernels_$TAG.json \
    > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?; echo "bench exit $rc"
cut -c1-400 gpurun_out/bench_$TAG.json
tail -4 gpurun_out/bench_$TAG.err
[ $rc -ne 0 ] && exit 1
cd /tmp
DZ_PROF_EVERY=1 timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- \
    python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32 --no-host-pass --kernel-table $REPO/gpurun_out/kernels_${TAG}_same_run.json > $REPO/gpurun_out/prof_$TAG.log 2>&1
rc=$?; echo "rocprof exit $rc"
cd $REPO
find gpurun_out/prof_$TAG -name '*kernel_trace*' -size +2M -delete
ls -R gpurun_out/prof_$TAG | head
[ $rc -ne 0 ] && exit 1
if bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1; then
  cp gpurun_out/traffic_$TAG.json profiles/traffic.json
  cp gpurun_out/mfma_$TAG.json profiles/mfma_util.json
else
  echo "PMC passes failed: keeping the committed profiles/traffic.json"
fi
tail -12 gpurun_out/pmc_$TAG.txt
