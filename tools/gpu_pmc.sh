#!/bin/bash
# HBM traffic of every kernel from the TCC counters, collected as MI355X_MICROARCH.md §HBM
# prescribes: separate --pmc passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), no
# trace domains besides --kernel-trace.  usage: tools/gpu_pmc.sh <tag>
TAG=${1:-r1}
REPO=$PWD
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$C -o pmc -- \
      python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-host-pass > $REPO/gpurun_out/pmc_${TAG}_$C.log 2>&1
  rc=$?; echo "$C exit $rc"
  [ $rc -ne 0 ] && { tail -5 $REPO/gpurun_out/pmc_${TAG}_$C.log; exit 1; }   # a faulting box: do not burn the budget
done
# matrix-core busy cycles (north-star: "MFMA utilisation against gfx950 peak"), own pass
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d $REPO/gpurun_out/pmc_${TAG}_MFMA -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-host-pass \
    > $REPO/gpurun_out/pmc_${TAG}_MFMA.log 2>&1
echo "MFMA exit $?"
cd $REPO
python tools/mfma_summary.py gpurun_out/pmc_${TAG}_MFMA gpurun_out/mfma_${TAG}.json | tail -14
ls -la gpurun_out/pmc_${TAG}_*/ | head -20
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/traffic_${TAG}.json
