#!/bin/bash
# SQ / LDS counters of ISOLATED kernels (tools/kbench.py), one rocprofv3 --pmc pass per counter group (--kernel-trace
# only, as MI355X_MICROARCH.md prescribes), for one or more environment settings:
#   tools/pmc_probe.sh <tag> "<kbench --only list>" ["ENV1=a ENV2=b" "ENV1=c" ...]
# -> gpurun_out/<tag>/pmc_probe.json: {setting: {kernel: {counter: mean per launch}}}
TAG=${1:?tag}; ONLY=${2:?kbench names}; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
[ $# -eq 0 ] && set -- "X=1"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $REPO/$OUT/sq_counters_available.txt
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM")
s=0
for SETTING in "$@"; do
  s=$((s+1)); i=0
  for G in "${GROUPS_[@]}"; do
    i=$((i+1)); C=""
    for c in $G; do grep -qx "$c" $REPO/$OUT/sq_counters_available.txt && C="$C $c"; done
    [ -z "$C" ] && continue
    env $SETTING timeout -s KILL 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/$OUT/pmc_${s}_$i -o pmc -- \
        python $REPO/tools/kbench.py --only $ONLY --reps 5 > $REPO/$OUT/pmc_${s}_$i.log 2>&1
    echo "setting $s [$SETTING] pass $i ($C) exit $?"
  done
done
cd $REPO
python - "$OUT" "$@" <<'PY'
import csv, glob, collections, json, sys
out_dir, settings = sys.argv[1], sys.argv[2:]
res = {}
for s, name in enumerate(settings, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{out_dir}/pmc_{s}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"][:70]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    res[name] = {}
    for k, cs in sorted(acc.items()):
        if any(x in k for x in ("at::", "rocclr", "elementwise", "distribution", "fill")):
            continue
        res[name][k] = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
        print(name, "|", k, res[name][k])
json.dump(res, open(f"{out_dir}/pmc_probe.json", "w"), indent=1)
PY
find $OUT/pmc_* -name '*kernel_trace*' -delete 2>/dev/null
find $OUT/pmc_* -name '*counter_collection.csv' -size +2M -delete 2>/dev/null
