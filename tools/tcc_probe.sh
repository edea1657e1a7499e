#!/bin/bash
# TCC (L2) counters of isolated kernels (tools/kbench.py), one --pmc pass per counter group: what the L2 asks of the
# fabric (read / write requests and their sizes), its hit rate, and FETCH_SIZE / WRITE_SIZE for the same launches.
#   usage: tools/tcc_probe.sh <tag> "<kbench --only list>"
TAG=${1:-r04_e}
ONLY=${2:-wave_stats,conv1_pool,conv2_pool}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCP_TCC_[A-Z0-9_]*" | sort -u > $REPO/gpurun_out/tcc_${TAG}_counters_available.txt
wc -l $REPO/gpurun_out/tcc_${TAG}_counters_available.txt
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_WRITE_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/tcc_${TAG}_$i -o pmc -- \
      python $REPO/tools/kbench.py --only $ONLY --reps 5 > $REPO/gpurun_out/tcc_${TAG}_$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $REPO
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/tcc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"][:90]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in sorted(acc.items()):
    if "at::" in k or "rocclr" in k or "elementwise" in k or "distribution" in k:
        continue
    out[k] = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
    out[k]["launches"] = max(v[1] for v in cs.values())
    print(k, out[k])
json.dump(out, open("gpurun_out/tcc_${TAG}.json", "w"), indent=1)
PY
find gpurun_out/tcc_${TAG}_* -name '*kernel_trace*' -delete 2>/dev/null
find gpurun_out/tcc_${TAG}_* -name '*counter_collection.csv' -size +2M -delete 2>/dev/null
