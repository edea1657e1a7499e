#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for i in 1 2 3; do
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc off --no-cpu-baseline --no-rehearsal > gpurun_out/bench_r4y_$i.json 2> gpurun_out/bench_r4y_$i.err
grep "timed region\|exact-f32 pass\|host-fed" gpurun_out/bench_r4y_$i.err | cut -c1-220
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/bench_r4y_$i.json") if l.startswith('{"metric"')][-1])
print("value", d["value"], d["ms_per_step"], d["host"]["step_period_ms_in_timed_region"], "| exact", d["exact_f32"]["value"], d["exact_f32"]["ms_per_step"], d["exact_f32"]["host_step_period_ms"])
PY
done
echo "=== exact f32 as the main pass, 20 steps and 100 steps"
for K in 20 100; do
timeout -s KILL 300 python bench.py --steps $K --warmup 5 --precision f32 --pmc off --no-cpu-baseline --no-rehearsal --no-host-pass 2> /dev/null | grep '^{"metric"' | cut -c1-250
done
