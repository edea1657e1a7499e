#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm_pre" -p no:cacheprovider 2>&1 | tail -3
timeout 500 python tools/g2bench.py --out gpurun_out/g2bench_r4j.json 2>&1 | grep -v amdgpu.ids | cut -c1-60 | tail -8
timeout 300 python tools/g2ablate.py --out gpurun_out/g2ablate_r4j.json 2>&1 | grep -v amdgpu.ids | tail -9
