#!/bin/bash
# visit U: k_gemm_f32.hip — parity (new test + the f32 tests that now route through it), isolated timing against the
# round-1 kernel, the exact-f32 step
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== kernel tests"
timeout -s KILL 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "gemm_f32 or convgemm or reflect" 2>&1 | grep -v amdgpu.ids | tail -15
echo "=== isolated"
timeout -s KILL 300 python tools/kbench.py --only lstm_proj,seg_mlp0,tdnn2,tdnn4,tdnn5,tdnn2_flat,tdnn5_flat 2>&1 | grep -v amdgpu.ids | grep " us \|vs" | cut -c1-110
cp gpurun_out/kbench.json gpurun_out/kbench_r4u.json
echo "=== model / pipeline tests on the f32 path"
timeout -s KILL 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_r2.py tests/test_gpu_der.py tests/test_gpu_ecapa.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -6
echo "=== exact-f32 step: new kernel, then the round-1 kernel"
for E in 1 0; do
DZ_F32_GEMM=$E timeout -s KILL 300 python bench.py --steps 100 --warmup 10 --precision f32 --pmc off --no-cpu-baseline --no-rehearsal --no-host-pass 2> /dev/null | grep '^{"metric"' | cut -c1-250
done
