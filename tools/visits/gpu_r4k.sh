#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_long_horizon.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -60 | cut -c1-330 | tee gpurun_out/long_horizon_r4k.txt
