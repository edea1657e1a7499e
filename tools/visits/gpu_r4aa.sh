#!/bin/bash
# visit AA: throughput against the number of streams per step (configs[1] is 64; the others are context)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for N in 16 32 64 128 256; do
timeout -s KILL 300 python bench.py --streams $N --steps 100 --warmup 10 --pmc off --no-cpu-baseline --no-rehearsal --no-exact-f32 --no-host-pass 2> gpurun_out/streams_$N.err | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($N, 'streams/step:', d['value'], 'xRT', d['ms_per_step'], 'ms/step', d['config'].get('streams_per_gpu'))"
tail -2 gpurun_out/streams_$N.err | cut -c1-200
done
