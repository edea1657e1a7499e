#!/bin/bash
# after the output tail's buffers moved in front of the first launch + warm steps: traced runs (stall hunt), two-rank rehearsal
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
cd /tmp
for i in 1 2 3 4 5 6; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/stall_r03_f_$i -o prof -- \
    python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-pass --pmc off > /dev/null 2>&1
done
cd $REPO
python tools/stall_report.py gpurun_out/stall_r03_f.json gpurun_out/stall_r03_f_[1-6] | cut -c1-260
rm -rf gpurun_out/stall_r03_f_[1-6]
DZ_FORCE_DEVICE=0 DZ_DIST_BACKEND=gloo timeout -s KILL 300 python bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-pass > gpurun_out/bench_r03_f_two_ranks.json 2> gpurun_out/bench_r03_f_two_ranks.err
echo "two ranks exit $?"; cut -c1-200 gpurun_out/bench_r03_f_two_ranks.json
timeout 200 python tools/benchmark_files.py --files 16 --seconds 600 --ami-hparams --workdir gpurun_out/bf16 2>&1 | tail -1 | cut -c1-300
rm -rf gpurun_out/bf16
