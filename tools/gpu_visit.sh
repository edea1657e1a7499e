#!/bin/bash
# One GPU-box visit, made of named sections (run from the repository root on the box):
#   tools/gpu_visit.sh <tag> <section> [<section> ...]
# Outputs go to gpurun_out/<tag>/ (merged back by gpurun); `tools/gpu_visit.sh <tag> collect` afterwards, run
# LOCALLY, copies the judged summaries into profiles/<tag>_*.
# sections: tests smoke driver driver2 long rocprof rocprof_serial config3 config5 config1 ranks8 ranks2 yardstick kbench files exp_tests
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT profiles
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
REPO=$PWD
last_json() { grep '^{"metric"' "$1" | tail -1; }
for SEC in "$@"; do
echo "=== $SEC"
case $SEC in
tests)
  timeout -s KILL 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -v amdgpu.ids > $OUT/pytest_gpu.txt
  tail -4 $OUT/pytest_gpu.txt
  grep -E "StreamBatch|Benchmark latency|DER vs the reference" $OUT/pytest_gpu.txt | cut -c1-330 > $OUT/long_horizon.txt ;;
exp_tests)   # the experiments build: the never-default kernels against the same gates
  DZ_EXPERIMENTS=1 timeout -s KILL 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_r2.py tests/test_abi.py -q -m "gpu or not gpu" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/pytest_gpu_experiments.txt ;;
smoke)
  timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
driver)      # exactly what the driver runs
  DZ_PMC_KEEP=$OUT/pmc_tables timeout -s KILL 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --pmc all --details $OUT/bench_driver_details.json > $OUT/bench_driver.json 2> $OUT/bench_driver.err
  echo "exit $? chars $(wc -c < $OUT/bench_driver.json)"; cut -c1-260 $OUT/bench_driver.json ;;
driver2)     # twice more, short form (no PMC children, no CPU leg): run-to-run spread of the 20-step region
  for i in 1 2; do
    timeout -s KILL 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --pmc off --no-cpu-baseline --no-rehearsal --details $OUT/bench_driver_short_${i}_details.json > $OUT/bench_driver_short_$i.json 2> $OUT/bench_driver_short_$i.err
    cut -c1-200 $OUT/bench_driver_short_$i.json
  done ;;
long)        # 200 timed steps
  timeout -s KILL 600 python bench.py --steps 200 --warmup 10 --pmc off --no-cpu-baseline --no-rehearsal --details $OUT/bench_200_details.json > $OUT/bench_200.json 2> $OUT/bench_200.err
  echo "exit $?"; cut -c1-260 $OUT/bench_200.json ;;
rocprof)     # rocprofv3 --kernel-trace --stats of the bench command, both precisions
  cd /tmp
  for P in f16x3 f32; do
    DZ_PROF_EVERY=1 DZ_SETTLE_STEPS=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_$P -o prof -- \
      python $REPO/bench.py --steps 10 --warmup 3 --precision $P --no-cpu-baseline --no-exact-f32 --no-host-pass --no-rehearsal --pmc off \
      --kernel-table $REPO/$OUT/kernels_events_rocprof_run_$P.json --details $REPO/$OUT/bench_rocprof_${P}_details.json > $REPO/$OUT/prof_$P.log 2>&1
    echo "rocprof $P exit $?"
  done
  cd $REPO
  for P in f16x3 f32; do
    python tools/prof_summary.py $OUT/prof_$P $OUT/kernel_stats_$P.md "rocprofv3 --kernel-trace --stats, bench.py --steps 10 --precision $P ($TAG)" > /dev/null
    cp $(find $OUT/prof_$P -name '*kernel_stats.csv' | head -1) $OUT/rocprofv3_kernel_stats_$P.csv
  done
  find $OUT -name '*kernel_trace*' -size +3M -delete ;;
rocprof_serial)   # rocprofv3 --kernel-trace --stats of the SERIALISED roofline pass alone (one lane, one HIP stream): the per-kernel
             # durations bench.py's `roofline` is computed from (tests/test_roofline_repro.py holds the line to this csv)
  cd /tmp
  for P in f16x3 f32; do
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_serial_$P -o prof -- \
      python $REPO/bench.py --serial-only --serial-steps 10 --precision $P --pmc off --details $REPO/$OUT/bench_serial_${P}_details.json \
      > $REPO/$OUT/bench_serial_$P.json 2> $REPO/$OUT/prof_serial_$P.log
    echo "rocprof serial $P exit $?"
  done
  cd $REPO
  for P in f16x3 f32; do
    python tools/prof_summary.py $OUT/prof_serial_$P $OUT/kernel_stats_serial_$P.md "rocprofv3 --kernel-trace --stats, bench.py --serial-only --serial-steps 10 --precision $P ($TAG)" > /dev/null
    cp $(find $OUT/prof_serial_$P -name '*kernel_stats.csv' | head -1) $OUT/rocprofv3_kernel_stats_serial_$P.csv
  done
  find $OUT -name '*kernel_trace*' -size +3M -delete ;;
config3)
  timeout -s KILL 300 python bench.py --config 3 --steps 20 --warmup 3 --details $OUT/bench_config3_details.json > $OUT/bench_config3.json 2> $OUT/bench_config3.err
  cut -c1-300 $OUT/bench_config3.json; grep "phases" $OUT/bench_config3.err | cut -c1-300 ;;
config5)
  timeout -s KILL 300 python bench.py --config 5 --steps 400 --warmup 20 --details $OUT/bench_config5_details.json > $OUT/bench_config5.json 2> $OUT/bench_config5.err
  cut -c1-1200 $OUT/bench_config5.json ;;
config1)
  timeout -s KILL 300 python bench.py --config 1 --steps 10 --warmup 2 --details $OUT/bench_config1_details.json > $OUT/bench_config1.json 2> $OUT/bench_config1.err
  cut -c1-1200 $OUT/bench_config1.json ;;
ranks2|ranks8)   # the multi-rank path of bench.py as ONE command; on a one-GPU box every rank shares the GPU (gloo rehearsal)
  N=${SEC#ranks}
  NG=$(python -c "import torch;print(torch.cuda.device_count())")
  # (rehearsal: N ranks on ONE GPU — each keeps the two-lane engine, N x 6 lanes of the 64-stream default would only queue)
  if [ "$NG" -ge "$N" ]; then ENVX="X=1"; EARGS=""; else ENVX="DZ_FORCE_DEVICE=0 DZ_DIST_BACKEND=gloo"; EARGS="--lanes 2 --recurrence valu"; fi
  env $ENVX timeout -s KILL 600 python bench.py --gpus $N $EARGS --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32 --no-host-pass --no-rehearsal --pmc off --details $OUT/bench_${N}_ranks_details.json > $OUT/bench_${N}_ranks.json 2> $OUT/bench_${N}_ranks.err
  echo "exit $? lines $(grep -c '^{' $OUT/bench_${N}_ranks.json) chars $(wc -c < $OUT/bench_${N}_ranks.json)"; cut -c1-700 $OUT/bench_${N}_ranks.json
  grep -E "process group up|cpu affinity" $OUT/bench_${N}_ranks.err | cut -c1-200 | head -16 ;;
yardstick)
  timeout -s KILL 400 python tools/gemm_yardstick.py --out $OUT/gemm_yardstick.json 2>&1 | grep -v amdgpu.ids | cut -c1-1500 ;;
kbench)
  timeout -s KILL 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids | grep " us " | cut -c1-70 | head -70
  cp gpurun_out/kbench.json $OUT/kbench_isolated.json ;;
files)       # config 4 shape on one GPU (16 files, AMI hyper-parameters), and the one-file-at-a-time loop for comparison
  timeout -s KILL 300 python tools/benchmark_files.py --files 16 --seconds 600 --ami-hparams --workdir $OUT/bf16 2>&1 | tail -1 | cut -c1-400 | tee $OUT/file_benchmark_config4.json
  rm -rf $OUT/bf16 ;;
ktest)       # quick: the kernel tests of the files named in KTEST (default: conv0 tests), then the conv0 lines of kbench
  timeout -s KILL 600 python -m pytest ${KTEST:-tests/test_gpu_kernels.py -k conv0} -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -15
  timeout -s KILL 300 python tools/kbench.py --only ${KONLY:-sinc_conv0_split,sinc_conv0_pair} 2>&1 | grep -v amdgpu.ids | grep " us " | cut -c1-100 ;;
ab)          # same-visit A/B of bench.py (200 steps, headline pass only) over environment settings AB_A / AB_B and / or
             # bench arguments AB_ARGS_A / AB_ARGS_B, e.g. AB_ARGS_A="--recurrence 3 --lanes 6" AB_ARGS_B="--lanes 4";
             # alternating, ${AB_N:-2} rounds
  for i in $(seq 1 ${AB_N:-2}); do
    for arm in A B; do
      if [ $arm = A ]; then E="${AB_A:-X=1}"; XA="$AB_ARGS_A"; else E="${AB_B:-X=1}"; XA="$AB_ARGS_B"; fi
      env $E timeout -s KILL 300 python bench.py --steps ${AB_STEPS:-200} --warmup 10 --pmc off --no-cpu-baseline --no-rehearsal --no-exact-f32 --no-host-pass $XA \
        --details $OUT/ab_${arm}_${i}_details.json > $OUT/ab_${arm}_$i.json 2> $OUT/ab_${arm}_$i.err
      echo "$arm$i [$E $XA] $(python -c "import json,sys; d=json.load(open('$OUT/ab_${arm}_$i.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
    done
  done ;;
verify_dry)  # tools/verify_real.py end to end on a stand-in corpus (synthetic checkpoints + WAVs): twice, the second run
             # scored against the first one's RTTMs as "the reference's hypothesis" (must be 0 %)
  python tools/make_dry_corpus.py $OUT/dry 4 120 > /dev/null
  timeout -s KILL 600 python tools/verify_real.py --ckpt $OUT/dry/ckpt --ami $OUT/dry/ami --out $OUT/dry/run1 2>&1 | grep -v amdgpu.ids | grep "verify_real\|x_real_time\|der_vs" | cut -c1-200
  cat $OUT/dry/run1/rttm_latency0.5s/*.rttm > $OUT/dry/expected.rttm
  timeout -s KILL 600 python tools/verify_real.py --ckpt $OUT/dry/ckpt --ami $OUT/dry/ami --out $OUT/dry/run2 --skip-gates --expected $OUT/dry/expected.rttm 2>&1 | grep "der_vs_reference_hypothesis\|files_compared" | cut -c1-200
  cp $OUT/dry/run2/verify_real.json $OUT/verify_real_dry_run.json; rm -rf $OUT/dry ;;
collect)     # LOCAL: judged copies
  for f in bench_8_ranks.json bench_2_ranks.json; do      # (gloo prints its own lines on stdout in the rehearsal: keep the bench line)
    [ -f $OUT/$f ] && last_json $OUT/$f > $OUT/$f.tmp && mv $OUT/$f.tmp $OUT/$f
  done
  for f in bench_driver.json bench_driver_details.json bench_200.json kernels_events_bench_run.json kernel_stats_f16x3.md kernel_stats_f32.md \
           rocprofv3_kernel_stats_f16x3.csv rocprofv3_kernel_stats_f32.csv kernels_events_rocprof_run_f16x3.json kernels_events_rocprof_run_f32.json \
           rocprofv3_kernel_stats_serial_f16x3.csv rocprofv3_kernel_stats_serial_f32.csv kernel_stats_serial_f16x3.md kernel_stats_serial_f32.md \
           bench_serial_f16x3.json bench_serial_f32.json \
           bench_config3.json bench_config5.json bench_config1.json bench_8_ranks.json bench_2_ranks.json gemm_yardstick.json kbench_isolated.json \
           long_horizon.txt file_benchmark_config4.json pytest_gpu_experiments.txt bench_driver_short_1.json bench_driver_short_2.json verify_real_dry_run.json; do
    [ -f $OUT/$f ] && cp $OUT/$f profiles/${TAG}_$f
  done
  [ -f $OUT/pytest_gpu.txt ] && tail -3 $OUT/pytest_gpu.txt > profiles/${TAG}_pytest_gpu_tail.txt
  for f in traffic.json traffic_f32.json mfma_util.json mfma_util_f32.json; do [ -f $OUT/pmc_tables/$f ] && cp $OUT/pmc_tables/$f profiles/$f; done
  ls profiles | grep "^${TAG}_" ;;
*) echo "unknown section $SEC" ;;
esac
done
