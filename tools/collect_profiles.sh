#!/bin/bash
# Run LOCALLY after a tools/gpu_final_r3.sh / gpu_final_r4.sh visit: gpurun merges only gpurun_out/ back, so the judged
# copies under profiles/ are made here from the merged files.   usage: tools/collect_profiles.sh <tag>
TAG=${1:-r03_a}
for P in f16x3 f32; do
  python tools/prof_summary.py gpurun_out/prof_${TAG}_$P profiles/${TAG}_kernel_stats_$P.md "rocprofv3 --kernel-trace --stats, bench.py --steps 10 --warmup 3 --precision $P, every launch instrumented ($TAG)" > /dev/null
  cp gpurun_out/prof_${TAG}_$P/prof_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats_$P.csv
done
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
[ -f gpurun_out/latency.json ] && cp gpurun_out/latency.json profiles/${TAG}_latency_config5.json
[ -f gpurun_out/g2bench_${TAG}.json ] && cp gpurun_out/g2bench_${TAG}.json profiles/${TAG}_gemm_generations_isolated.json
[ -f gpurun_out/long_horizon_${TAG}.txt ] && cp gpurun_out/long_horizon_${TAG}.txt profiles/${TAG}_long_horizon.txt
[ -f gpurun_out/pytest_gpu_${TAG}.txt ] && tail -3 gpurun_out/pytest_gpu_${TAG}.txt > profiles/${TAG}_pytest_gpu_tail.txt
ls profiles | grep ${TAG}
