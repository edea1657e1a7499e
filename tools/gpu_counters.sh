#!/bin/bash
# SQ / TCC counters of the isolated kernels (tools/kbench.py) in separate --pmc passes.
#   usage: tools/gpu_counters.sh <tag> "<kbench --only list>"
TAG=${1:-r02_c}
ONLY=${2:-tdnn2_pre,lstm_proj_pre,conv1_pool,conv2_pool,sinc_conv0,lstm,mlp_head}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
timeout -s KILL 120 python tools/kbench.py --only $ONLY > gpurun_out/cnt_${TAG}_plain.log 2>&1   # a plain GPU process first
cd /tmp
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/cnt_${TAG}_$i -o pmc -- \
      python $REPO/tools/kbench.py --only $ONLY > $REPO/gpurun_out/cnt_${TAG}_$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $REPO
python - <<PY
import csv, glob, collections
for i in (1, 2, 3):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob("gpurun_out/cnt_${TAG}_%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"][:70]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, cs in sorted(acc.items()):
        if "at::" in k or "rocclr" in k or "elementwise" in k or "distribution" in k:
            continue
        print(k, {c: round(v[0] / v[1]) for c, v in cs.items()}, "launches", max(v[1] for v in cs.values()))
PY
find gpurun_out/cnt_${TAG}_* -name '*kernel_trace*' -size +1M -delete 2>/dev/null
