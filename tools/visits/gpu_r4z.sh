#!/bin/bash
# visit Z: matrix-pipe busy fraction and the LDS counters of the three pre-split GEMM generations ALONE
# (review item 1's "done" line asks for PMC MFMA-busy isolated with SQ_LDS_* next to it)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/sqg_r4z_$i -o pmc -- \
      python $REPO/tools/g2bench.py --no-rec --reps 4 --only tdnn2,proj --out $REPO/gpurun_out/g2bench_r4z.json > $REPO/gpurun_out/sqg_r4z_$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/sqg_r4z_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = r["Kernel_Name"][:52].replace("void (anonymous namespace)::", "") + " grid=" + r.get("Grid_Size", "?")
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in sorted(acc.items()):
    if "gemm" not in k:
        continue
    o = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
    if o.get("GRBM_GUI_ACTIVE"):
        o["mfma_busy_frac"] = round(o.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (o["GRBM_GUI_ACTIVE"] / 8), 3)   # per SIMD / per-XCD cycles
    out[k] = o
    print(k, o)
json.dump(out, open("gpurun_out/sqg_r4z.json", "w"), indent=1)
PY
find gpurun_out/sqg_r4z_* -name '*kernel_trace*' -delete 2>/dev/null
find gpurun_out/sqg_r4z_* -name '*counter_collection.csv' -size +2M -delete 2>/dev/null
