#!/bin/bash
# visit T: HBM-side traffic of the pre-split GEMM generations ALONE (FETCH_SIZE / WRITE_SIZE / L2 hit rate per kernel),
# to split the in-pipeline excess (gemm_pre<3> 121 MB vs 79 algorithmic) into "the kernel's own" and "contention"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/tccg_r4t_$i -o pmc -- \
      python $REPO/tools/g2bench.py --no-rec --reps 4 --only tdnn2,tdnn4,tdnn5,proj --out $REPO/gpurun_out/g2bench_r4t.json > $REPO/gpurun_out/tccg_r4t_$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/tccg_r4t_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        # rows of one kernel differ by layer: key on the grid size too
        key = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in sorted(acc.items()):
    if "at::" in k or "rocclr" in k or "elementwise" in k or "distribution" in k or "gemm" not in k:
        continue
    out[k] = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
    out[k]["launches"] = max(v[1] for v in cs.values())
    f, w = out[k].get("FETCH_SIZE", 0), out[k].get("WRITE_SIZE", 0)
    out[k]["hbm_mb"] = round((2 * f + w) * 1024 / 1e6, 1)
    print(k, out[k])
json.dump(out, open("gpurun_out/tccg_r4t.json", "w"), indent=1)
PY
find gpurun_out/tccg_r4t_* -name '*kernel_trace*' -delete 2>/dev/null
find gpurun_out/tccg_r4t_* -name '*counter_collection.csv' -size +2M -delete 2>/dev/null
