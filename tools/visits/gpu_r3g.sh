#!/bin/bash
# Round 3, visit g: statistics pooling fused into tdnn5 (DZ_POOL_FUSE)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== tests"
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_r2.py tests/test_gpu_der.py tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12
SKIP_TESTS=1 NK=13 bash tools/gpu_ab.sh r3g none "DZ_POOL_FUSE=1 DZ_POOL_FUSE=0 DZ_POOL_FUSE=1 DZ_POOL_FUSE=0" | cut -c1-150
