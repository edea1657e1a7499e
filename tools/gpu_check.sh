#!/bin/bash
# Run every GPU test function in its own process (a GPU memory fault kills the process),
# append a one-line verdict per function and keep the tails in gpurun_out/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/gpu_check.log
: > $OUT
(rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -6; rocm-smi --showmeminfo vram | head -8) > gpurun_out/hw.txt 2>&1
FILES=${@:-tests/test_gpu_kernels.py tests/test_gpu_models.py}
for f in $FILES; do
  for fn in $(grep -oE "^def (test_[a-z0-9_]+)" $f | awk '{print $2}'); do
    echo "=== $f::$fn" >> $OUT
    timeout 900 python -m pytest "$f::$fn" -m gpu -q -s --timeout 600 -p no:cacheprovider 2>&1 | tail -40 >> $OUT
    echo "--- exit ${PIPESTATUS[0]}" >> $OUT
  done
done
grep -E "^===|passed|failed|error|exit" $OUT | paste - - - | cut -c1-220
