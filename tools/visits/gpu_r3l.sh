#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_r2.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
SKIP_TESTS=1 NK=8 bash tools/gpu_ab.sh r3l none "DZ_PRIO_A=0 DZ_PRIO_A=-1 DZ_PRIO_A=0 DZ_PRIO_A=-1 DZ_PRIO_B=-1 GPU_MAX_HW_QUEUES=6 GPU_MAX_HW_QUEUES=12" | cut -c1-150
