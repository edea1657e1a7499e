#!/bin/bash
# visit W: where the new exact-f32 GEMM's time goes: matrix-pipe busy cycles, clock, wave-cycle buckets (isolated launches)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $REPO/gpurun_out/sq_counters_available.txt
wc -l $REPO/gpurun_out/sq_counters_available.txt
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/sqf_r4w_$i -o pmc -- \
      python $REPO/tools/kbench.py --only tdnn5_flat,lstm_proj --reps 5 > $REPO/gpurun_out/sqf_r4w_$i.log 2>&1
  echo "pass $i ($C) exit $?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/sqf_r4w_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = r["Kernel_Name"][:58] + " grid=" + r.get("Grid_Size", "?")
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob("gpurun_out/sqf_r4w_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = r["Kernel_Name"][:58] + " grid=" + str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))) if "Grid_Size_X" in r else r["Kernel_Name"][:58]
        d = dur[key]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
out = {}
for k, cs in sorted(acc.items()):
    if "gemm_f32" not in k and "convgemm" not in k:
        continue
    out[k] = {c: round(v[0] / v[1], 1) for c, v in cs.items()}
    print(k, out[k])
for k, d in sorted(dur.items()):
    if "gemm_f32" in k or "convgemm" in k:
        print("duration us under the counter pass", k, round(d[0] / d[1], 1), d[1])
json.dump(out, open("gpurun_out/sqf_r4w.json", "w"), indent=1)
PY
find gpurun_out/sqf_r4w_* -name '*kernel_trace*' -size +1M -delete 2>/dev/null
find gpurun_out/sqf_r4w_* -name '*counter_collection.csv' -size +2M -delete 2>/dev/null
