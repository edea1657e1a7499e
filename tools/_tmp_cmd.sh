mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; REPO=$PWD
python tools/kbench.py --only lstm_proj,tdnn2,tdnn5 2>&1 | grep -v "^    \|amdgpu.ids"
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_bx3 -o pmc -- python $REPO/tools/kbench.py --only tdnn2 --reps 3 > $REPO/gpurun_out/pmc_bx3.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for f in glob.glob('gpurun_out/pmc_bx3/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,c in acc.items():
    if 'gemm' in k:
        w=c.get('SQ_WAVE_CYCLES',1)
        print(k); print({a: round(b/w,3) for a,b in c.items()})
PY
