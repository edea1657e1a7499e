#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 300 python tools/g2ablate.py --out gpurun_out/g2ablate_r4i.json 2>&1 | grep -v amdgpu.ids | tail -5
