#!/bin/bash
# Round 3, visit a: everything new on the host side — full GPU suite, the driver-form bench with the
# live PMC passes and the exact-f32 roofline, bench.py --gpus 2 as ONE process, the batched file
# benchmark next to the one-file-at-a-time loop.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
T=r03_a
echo "=== pytest -m gpu"
timeout -s KILL 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15
echo "=== bench (driver form)"
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${T}_driver.json 2> gpurun_out/bench_${T}_driver.err
echo "exit $?"; cut -c1-300 gpurun_out/bench_${T}_driver.json; grep -i "pmc\|host threads\|exact" gpurun_out/bench_${T}_driver.err | cut -c1-250
echo "=== bench 200 steps"
timeout -s KILL 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --pmc off > gpurun_out/bench_${T}.json 2> gpurun_out/bench_${T}.err
echo "exit $?"; cut -c1-200 gpurun_out/bench_${T}.json
echo "=== file benchmark: batched vs loop (config-4 shape, 8 files)"
timeout -s KILL 300 python tools/benchmark_files.py --files 8 --seconds 300 --ami-hparams --workdir gpurun_out/bf_batched 2>&1 | tail -1 | cut -c1-400
DZ_CONCURRENT_FILES=0 timeout -s KILL 300 python tools/benchmark_files.py --files 8 --seconds 300 --ami-hparams --workdir gpurun_out/bf_loop 2>&1 | tail -1 | cut -c1-400
for f in gpurun_out/bf_batched/rttm_w1/*.rttm; do cmp -s $f gpurun_out/bf_loop/rttm_w1/$(basename $f) || echo "DIFF $f"; done; echo "rttm compared"
timeout -s KILL 120 python tools/benchmark_files.py --files 1 --seconds 30 --workdir gpurun_out/bf_c1 2>&1 | tail -1 | cut -c1-300
rm -rf gpurun_out/bf_batched/wav gpurun_out/bf_loop/wav gpurun_out/bf_c1/wav
