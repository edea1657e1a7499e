#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== full GPU suite"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12
echo "=== range map"
cat gpurun_out/f16x3_range.json | python -c "
import json,sys
for r in json.load(sys.stdin):
    print(r)
" | head -50
echo "=== bench driver form"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc all > gpurun_out/bench_r3h_driver.json 2> gpurun_out/bench_r3h_driver.err; echo "exit $?"; cut -c1-200 gpurun_out/bench_r3h_driver.json; grep -i "pmc\|host threads" gpurun_out/bench_r3h_driver.err | cut -c1-200
