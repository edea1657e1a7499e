#!/bin/bash
# Round 3, visit c: gemm_pre interleaved k-loop (DZ_GP_LOOP) — isolated kernels, parity tests, pipeline A/B
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for v in 0 1; do
  echo "=== kbench DZ_GP_LOOP=$v"
  DZ_GP_LOOP=$v timeout 200 python tools/kbench.py --only tdnn2,tdnn3,tdnn4,tdnn5,lstm_proj,seg_mlp0 2>&1 | grep -v amdgpu.ids | grep "_pre" | cut -c1-100
done
echo "=== tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_r2.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
SKIP_TESTS=1 bash tools/gpu_ab.sh r3c none "DZ_GP_LOOP=1 DZ_GP_LOOP=0 DZ_GP_LOOP=1 DZ_GP_LOOP=0" | grep -v "^   " | cut -c1-160
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open(f"gpurun_out/bench_r3c_{i}.json"))
    print(i, d["value"], [(k["kernel"][:22], k["avg_launch_us"]) for k in d["roofline_kernels"][:8]])
PY
