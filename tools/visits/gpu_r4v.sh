#!/bin/bash
# visit V: k_gemm_f32.hip with four 16-wide stages and counted vmcnt
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== kernel tests"
timeout -s KILL 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "gemm_f32 or convgemm or reflect" 2>&1 | grep -v amdgpu.ids | tail -5
echo "=== isolated"
timeout -s KILL 300 python tools/kbench.py --only lstm_proj,seg_mlp0,tdnn4,tdnn2_flat,tdnn5_flat 2>&1 | grep -v amdgpu.ids | grep " us " | grep -v "split\|pre" | cut -c1-110
cp gpurun_out/kbench.json gpurun_out/kbench_r4v.json
echo "=== exact-f32 step: new kernel, then the round-1 kernel"
for E in 1 0; do
DZ_F32_GEMM=$E timeout -s KILL 300 python bench.py --steps 100 --warmup 10 --precision f32 --pmc off --no-cpu-baseline --no-rehearsal --no-host-pass 2> /dev/null | grep '^{"metric"' | cut -c1-250
done
