#!/bin/bash
# marginal cost of each launch class inside the overlapped pipeline (DZ_SKIP, api.hip)
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/skip_${1:-a}.log; : > $OUT
for m in 0 1 2 4 8 16 32 64 128 256 512 0 1023; do
  DZ_SKIP=$m DZ_NO_PROF=1 timeout 120 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-host-pass 2>&1 >/dev/null | grep "timed region" | sed "s/^/skip=$m /" | cut -c1-120 >> $OUT
done
cat $OUT
