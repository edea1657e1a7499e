#!/bin/bash
# stall hunt after the up-front arena allocation: six traced runs
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
cd /tmp
for i in 1 2 3 4 5 6; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/stall_b_$i -o prof -- \
    python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-pass --pmc off > /dev/null 2>&1
done
cd $REPO
python tools/stall_report.py gpurun_out/stall_r03_b.json gpurun_out/stall_b_[1-6] | cut -c1-300
rm -rf gpurun_out/stall_b_[1-6]
SKIP_TESTS=1 NK=3 bash tools/gpu_ab.sh r3n none "DZ_GEMM_TAIL=0 DZ_GEMM_TAIL=1 DZ_GEMM_TAIL=0 DZ_GEMM_TAIL=1" | grep -v "^   \|^\[bench" | cut -c1-150
