#!/bin/bash
# schedule of the step from the dispatches' own timestamps (tools/timeline.py), three configurations
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp DZ_PROF_EVERY=1
i=0
for cfg in "DZ_SEG_FRONT=0,DZ_INFLIGHT=3" "DZ_SEG_FRONT=1,DZ_INFLIGHT=4" "DZ_SEG_FRONT=0,DZ_INFLIGHT=5" "DZ_SEG_FRONT=1,DZ_INFLIGHT=3"; do
  i=$((i+1))
  rm -f gpurun_out/tl_$i.txt
  env $(echo $cfg | tr ',' ' ') DZ_PROF_TIMELINE=gpurun_out/tl_$i.txt timeout 120 python bench.py --steps 60 --warmup 10 \
      --no-cpu-baseline --no-exact-f32 --no-host-pass --pmc off > gpurun_out/tl_$i.json 2> gpurun_out/tl_$i.err
  echo "=== $cfg"; python -c "import json;d=json.load(open('gpurun_out/tl_$i.json'));print(d['ms_per_step'])"
  python tools/timeline.py gpurun_out/tl_$i.txt --steps 30:42 | cut -c1-230
done
