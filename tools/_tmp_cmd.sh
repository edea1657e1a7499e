mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python tools/kbench.py --only lstm_proj,tdnn2,tdnn5,tdnn1 2>&1 | grep -v "amdgpu.ids"
python -m pytest tests/test_gpu_models.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "forward" 2>&1 | grep -v "^$" | tail -12
for pr in f16x3; do for s in 1; do echo "== precision=$pr split=$s"; DZ_PRECISION=$pr DZ_SEG_SPLIT=$s python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "timed region"; done; done
