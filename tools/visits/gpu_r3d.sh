#!/bin/bash
# Round 3, visit d: conv_pool_h with batched tile loads; gemm_pre k-block-major addressing (timing only)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== kbench conv pools"
timeout 200 python tools/kbench.py --only conv1_pool,conv2_pool 2>&1 | grep -v amdgpu.ids | cut -c1-100
for v in 0 1; do
  echo "=== kbench DZ_GP_DBG=$v (1 = k-block-major addressing, WRONG results, timing only)"
  DZ_GP_DBG=$v timeout 200 python tools/kbench.py --only tdnn2,tdnn4,tdnn5,lstm_proj 2>&1 | grep -v amdgpu.ids | grep "_pre" | grep -v vs | cut -c1-100
done
echo "=== tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_r2.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
SKIP_TESTS=1 bash tools/gpu_ab.sh r3d none "DZ_GP_LOOP=1 DZ_GP_LOOP=1,DZ_GP_DBG=1 DZ_GP_LOOP=1 DZ_GP_LOOP=1,DZ_GP_DBG=1" | grep -v "^   " | grep -v "^\[bench" | cut -c1-160
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open(f"gpurun_out/bench_r3d_{i}.json"))
    print(i, d["value"], [(k["kernel"][:22], k["avg_launch_us"]) for k in d["roofline_kernels"][:10]])
PY
