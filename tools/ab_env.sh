#!/bin/bash
# same-visit A/B over environment settings: AB_ENVS="X=1;DZ_FOO=1;DZ_FOO=2", alternating, AB_N rounds, 20-step and 200-step forms
OUT=gpurun_out/${AB_TAG:-ab}; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
IFS=';' read -ra ENVS <<< "${AB_ENVS:-X=1}"
for rnd in $(seq 1 ${AB_N:-2}); do
  k=0
  for E in "${ENVS[@]}"; do
    k=$((k+1))
    for st in ${AB_FORMS:-20 200}; do
      W=5; [ $st != 20 ] && W=10
      env $E timeout -s KILL 300 python bench.py --steps $st --warmup $W --pmc off --no-cpu-baseline --no-rehearsal --no-exact-f32 --no-host-pass --serial-steps 0 $AB_ARGS --details $OUT/e${k}_${st}_$rnd.details.json > $OUT/e${k}_${st}_$rnd.json 2> $OUT/e${k}_${st}_$rnd.err
      echo "[$E] steps $st round $rnd: $(python -c "import json; d=json.load(open('$OUT/e${k}_${st}_$rnd.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
    done
  done
done
