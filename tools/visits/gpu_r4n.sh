#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for cfg in ${GRID}; do
  env $(echo $cfg | tr ',' ' ') timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact-f32 --pmc off --no-host-pass --no-rehearsal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], ' | '.join('%s %.0f' % (k['kernel'][:20], k['avg_launch_us']) for k in d['roofline_kernels'][:8]))"
done
