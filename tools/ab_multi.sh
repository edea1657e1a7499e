# same-visit multi-arm A/B of bench.py (200 steps, headline pass only, power entry on): ARMS="name:ENV=1,ENV2=x name2:..." ROUNDS=2
OUT=gpurun_out/${TAG:-ab_multi}; mkdir -p $OUT
for r in $(seq 1 ${ROUNDS:-2}); do
  for arm in $ARMS; do
    name=${arm%%:*}; envs=$(echo "${arm#*:}" | tr ',' ' ')
    env $envs timeout -s KILL 300 python bench.py --steps ${STEPS:-200} --warmup 10 --pmc off --no-cpu-baseline --no-rehearsal --no-exact-f32 --no-host-pass \
      --details $OUT/${name}_${r}_details.json > $OUT/${name}_$r.json 2> $OUT/${name}_$r.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/${name}_$r.json")); p=d.get("power") or {}
    print("$name r$r", d["value"], d["ms_per_step"], p.get("package_w"), p.get("sclk_mhz"), p.get("joules_per_step"))
except Exception as e:
    print("$name r$r FAILED", e)
PY
  done
done
