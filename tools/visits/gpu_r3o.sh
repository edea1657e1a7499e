#!/bin/bash
# keep the kernel trace of a run that shows the start-up stall
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
cd /tmp
for i in 1 2 3 4 5 6 7 8; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $REPO/gpurun_out/stall_c_$i -o prof -- \
    python $REPO/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-host-pass --pmc off > $REPO/gpurun_out/stall_c_$i.log 2>&1
done
cd $REPO
python tools/stall_report.py gpurun_out/stall_r03_c.json gpurun_out/stall_c_[1-8] | cut -c1-260
