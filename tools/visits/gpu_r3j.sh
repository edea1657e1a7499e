#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== full GPU suite"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
SKIP_TESTS=1 NK=14 bash tools/gpu_ab.sh r3j none "DZ_NORM_SPLIT=1 DZ_NORM_SPLIT=0 DZ_NORM_SPLIT=1 DZ_NORM_SPLIT=0" | cut -c1-150
echo "=== config 3"
timeout 300 python bench.py --config 3 --steps 20 --warmup 3 2>&1 | tail -2 | cut -c1-400
