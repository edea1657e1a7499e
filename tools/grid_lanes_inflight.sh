#!/bin/bash
# lanes x inflight grid, driver form (20 steps) and 200 steps, alternating, 2 rounds
OUT=gpurun_out/r06zb; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
for rnd in 1 2; do
for cfg in "6 8" "6 9" "6 10" "7 9" "7 10" "5 7" "8 10"; do
  set -- $cfg
  for st in 20 200; do
    W=5; [ $st = 200 ] && W=10
    timeout -s KILL 300 python bench.py --steps $st --warmup $W --lanes $1 --inflight $2 --pmc off --no-cpu-baseline --no-rehearsal --no-exact-f32 --no-host-pass --serial-steps 0 --details $OUT/g_$1_$2_${st}_$rnd.details.json > $OUT/g_$1_$2_${st}_$rnd.json 2> $OUT/g_$1_$2_${st}_$rnd.err
    echo "lanes $1 inflight $2 steps $st round $rnd: $(python -c "import json; d=json.load(open('$OUT/g_$1_$2_${st}_$rnd.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
  done
done
done
