#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 120 python tools/conv_pool_phases.py 2>&1 | grep -v amdgpu.ids | tail -40
echo "=== tests (norm-from-partials rewrite)"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_r2.py tests/test_gpu_der.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
SKIP_TESTS=1 NK=12 bash tools/gpu_ab.sh r3f none "DZ_GP_LOOP=1 DZ_GP_LOOP=1" | cut -c1-150
