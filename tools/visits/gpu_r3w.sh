#!/bin/bash
# config 3 after moving block 0 / Res2Net / asp_conv / DFT to the split-f16 kernel: parity tests, bench, kernel stats
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ecapa.py tests/test_gpu_der.py -m gpu -q -x --timeout 500 -p no:cacheprovider 2>&1 | tail -8
for i in 1 2; do
  timeout 200 python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/c3_$i.json 2> gpurun_out/c3_$i.err
  python -c "import json;d=json.load(open('gpurun_out/c3_$i.json'));print('config3', d['value'], d['ms_per_step'])"
done
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_c3 -o c3 -- \
  python $REPO/bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/c3_prof.json 2> $REPO/gpurun_out/c3_prof.err
cd $REPO
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_c3/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", tot / 1e6 / 12)
for r in rows[:16]:
    print("%-70s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
