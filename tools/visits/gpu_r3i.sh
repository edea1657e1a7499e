#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== full GPU suite"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
for v in 0 1 2 3; do
  echo "=== kbench DZ_GP_DBG=$v (1 k-block addressing, 2 no DMA, 3 both; timing only)"
  DZ_GP_DBG=$v timeout 200 python tools/kbench.py --only tdnn2,tdnn5,lstm_proj 2>&1 | grep -v amdgpu.ids | grep "_pre_f32out" | cut -c1-100
done
echo "=== bench driver form"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc all > gpurun_out/bench_r3i_driver.json 2> gpurun_out/bench_r3i_driver.err; echo "exit $?"; cut -c1-200 gpurun_out/bench_r3i_driver.json; grep -i "pmc\|Traceback\|Error" gpurun_out/bench_r3i_driver.err | cut -c1-200
