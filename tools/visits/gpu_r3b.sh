#!/bin/bash
# Round 3, visit b: VALU issue-rate microbenchmark (wall time, whole chip), the shared wave_stats A/B,
# the file benchmark with the background loader.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== fma_rate"
timeout 60 tools/ubench/fma_rate 2>&1 | tail -20
echo "=== tests"
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_der.py tests/test_reference_pipeline.py tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
echo "=== file benchmark"
timeout -s KILL 300 python tools/benchmark_files.py --files 8 --seconds 300 --ami-hparams --workdir gpurun_out/bf_batched 2>&1 | tail -1 | cut -c1-400
timeout -s KILL 300 python tools/benchmark_files.py --files 16 --seconds 600 --ami-hparams --workdir gpurun_out/bf_batched16 2>&1 | tail -1 | cut -c1-400
rm -rf gpurun_out/bf_batched/wav gpurun_out/bf_batched16
SKIP_TESTS=1 bash tools/gpu_ab.sh r3b none "DZ_SHARED_STATS=1 DZ_SHARED_STATS=0 DZ_SHARED_STATS=1 DZ_SHARED_STATS=0 DZ_DEPTH=3 DZ_DEPTH=3,DZ_PRIO_A=-1" | grep -v "^   " | cut -c1-160
