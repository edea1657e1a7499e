#!/bin/bash
# Round-4 end-of-round visit: everything the committed profiles/r03_<tag>_* come from.
#   usage: tools/gpu_final_r3.sh <tag>     (outputs under gpurun_out/, judged copies under profiles/)
TAG=${1:-r04_b}
mkdir -p gpurun_out profiles
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
echo "=== pytest -m gpu"
timeout -s KILL 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -v amdgpu.ids > gpurun_out/pytest_gpu_${TAG}.txt
tail -4 gpurun_out/pytest_gpu_${TAG}.txt
grep -E "StreamBatch|Benchmark latency|DER vs the reference" gpurun_out/pytest_gpu_${TAG}.txt | cut -c1-330 > gpurun_out/long_horizon_${TAG}.txt
wc -l gpurun_out/long_horizon_${TAG}.txt
echo "=== smoke"
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== bench, driver form (live PMC passes for both precisions, CPU baseline)"
timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc all > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err
echo "exit $?"; cut -c1-220 gpurun_out/bench_${TAG}_driver.json
echo "=== bench, 200 steps"
timeout -s KILL 600 python bench.py --steps 200 --warmup 10 --pmc off --no-cpu-baseline --no-rehearsal --kernel-table gpurun_out/kernels_${TAG}.json > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
echo "exit $?"; cut -c1-220 gpurun_out/bench_${TAG}.json
echo "=== rocprofv3 --kernel-trace --stats, both precisions"
cd /tmp
for P in f16x3 f32; do
  DZ_PROF_EVERY=1 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_$P -o prof -- \
    python $REPO/bench.py --steps 10 --warmup 3 --precision $P --no-cpu-baseline --no-exact-f32 --no-host-pass --no-rehearsal --pmc off \
    --kernel-table $REPO/gpurun_out/kernels_${TAG}_${P}_same_run.json > $REPO/gpurun_out/prof_${TAG}_$P.log 2>&1
  echo "rocprof $P exit $?"
done
echo "=== two more traced runs (stall check)"
for i in 1 2; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/stall_${TAG}_$i -o prof -- \
    python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-pass --no-rehearsal --pmc off > /dev/null 2>&1
done
cd $REPO
python tools/stall_report.py gpurun_out/stall_${TAG}.json gpurun_out/prof_${TAG}_f16x3 gpurun_out/stall_${TAG}_[1-2] | cut -c1-300
for P in f16x3 f32; do
  python tools/prof_summary.py gpurun_out/prof_${TAG}_$P profiles/${TAG}_kernel_stats_$P.md "rocprofv3 --kernel-trace --stats, bench.py --steps 10 --precision $P ($TAG)" > /dev/null
  cp $(find gpurun_out/prof_${TAG}_$P -name '*kernel_stats.csv' | head -1) profiles/${TAG}_rocprofv3_kernel_stats_$P.csv
done
find gpurun_out -name '*kernel_trace*' -size +3M -delete
rm -rf gpurun_out/stall_${TAG}_[1-2]
echo "=== isolated kernels"
timeout -s KILL 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids | grep " us " | cut -c1-70 | head -60
cp gpurun_out/kbench.json gpurun_out/kbench_${TAG}.json
echo "=== config 5 latency"
timeout -s KILL 200 python tools/latency.py 2>&1 | tail -3 | cut -c1-300
echo "=== file benchmark: config 1, config 4 shape (16 files), loop path for comparison"
timeout -s KILL 200 python tools/benchmark_files.py --files 1 --seconds 30 --workdir gpurun_out/bf1 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/bf_${TAG}_config1.json
timeout -s KILL 300 python tools/benchmark_files.py --files 16 --seconds 600 --ami-hparams --workdir gpurun_out/bf16 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/bf_${TAG}_config4.json
DZ_CONCURRENT_FILES=0 timeout -s KILL 300 python tools/benchmark_files.py --files 16 --seconds 600 --ami-hparams --workdir gpurun_out/bf16l 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/bf_${TAG}_config4_loop.json
for f in gpurun_out/bf16/rttm_w1/*.rttm; do cmp -s $f gpurun_out/bf16l/rttm_w1/$(basename $f) || echo "RTTM DIFF $f"; done; echo "rttm files compared"
rm -rf gpurun_out/bf1 gpurun_out/bf16 gpurun_out/bf16l
echo "=== config 3"
timeout -s KILL 300 python bench.py --config 3 --steps 20 --warmup 3 2> /dev/null | grep '^{"metric"' | tail -1 > gpurun_out/bench_${TAG}_config3.json
cut -c1-300 gpurun_out/bench_${TAG}_config3.json
echo "=== bench.py --gpus 2 as one process (rehearsal on this box's GPU count)"
NG=$(python -c "import torch;print(torch.cuda.device_count())")
if [ "$NG" -ge 2 ]; then ENVX=""; else ENVX="DZ_FORCE_DEVICE=0 DZ_DIST_BACKEND=gloo"; fi
env $ENVX timeout -s KILL 300 python bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-pass --no-rehearsal > gpurun_out/bench_${TAG}_two_ranks.json 2> gpurun_out/bench_${TAG}_two_ranks.err
echo "exit $?"; cut -c1-200 gpurun_out/bench_${TAG}_two_ranks.json; grep "process group up" gpurun_out/bench_${TAG}_two_ranks.err | cut -c1-160
echo "=== GEMM generations, isolated"
timeout -s KILL 400 python tools/g2bench.py --out gpurun_out/g2bench_${TAG}.json 2>&1 | grep -v amdgpu.ids | cut -c1-50 | tail -3
cp gpurun_out/g2bench_${TAG}.json profiles/${TAG}_gemm_generations_isolated.json
cp gpurun_out/long_horizon_${TAG}.txt profiles/${TAG}_long_horizon.txt
cp gpurun_out/pytest_gpu_${TAG}.txt gpurun_out/pytest_gpu_${TAG}.full.txt
tail -3 gpurun_out/pytest_gpu_${TAG}.txt > profiles/${TAG}_pytest_gpu_tail.txt
# judged copies
cp gpurun_out/bench_${TAG}_driver.json profiles/${TAG}_bench_driver_form.json
cp gpurun_out/bench_${TAG}.json profiles/${TAG}_bench.json
cp gpurun_out/kernels_${TAG}.json profiles/${TAG}_kernels_events_bench_run.json
cp gpurun_out/kernels_${TAG}_f16x3_same_run.json profiles/${TAG}_kernels_events_rocprof_run_f16x3.json
cp gpurun_out/kernels_${TAG}_f32_same_run.json profiles/${TAG}_kernels_events_rocprof_run_f32.json
cp gpurun_out/stall_${TAG}.json profiles/${TAG}_stall_report.json
cp gpurun_out/kbench_${TAG}.json profiles/${TAG}_kbench_isolated.json
cp gpurun_out/bench_${TAG}_config3.json profiles/${TAG}_bench_config3.json
grep "^{\"metric\"" gpurun_out/bench_${TAG}_two_ranks.json | tail -1 > profiles/${TAG}_bench_two_ranks_rehearsal.json
cat gpurun_out/bf_${TAG}_config1.json gpurun_out/bf_${TAG}_config4.json gpurun_out/bf_${TAG}_config4_loop.json > profiles/${TAG}_file_benchmark.jsonl
[ -f gpurun_out/f16x3_range.json ] && cp gpurun_out/f16x3_range.json profiles/${TAG}_f16x3_range_map.json
ls profiles | grep ${TAG}
