#!/bin/bash
# where does a config-3 step (powerset segmentation + ECAPA-TDNN, 32 windows = 96 embedding rows) spend its time?
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
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
for r in rows[:22]:
    print("%-70s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -2 gpurun_out/c3_prof.err | cut -c1-300
