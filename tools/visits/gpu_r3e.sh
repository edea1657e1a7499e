#!/bin/bash
# Round 3, visit e: MFMA dependent-chain distance; conv_pool time vs tiles per workgroup (batch sweep)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== mfma_dep"
timeout 60 tools/ubench/mfma_dep 2>&1 | tail -12
for b in 8 16 32 64 128; do
  echo "=== kbench --batch $b"
  timeout 200 python tools/kbench.py --batch $b --only conv1_pool,conv2_pool,sinc_conv0_split 2>&1 | grep -v amdgpu.ids | grep "convpool \|sinc_conv0_split" | cut -c1-100
done
