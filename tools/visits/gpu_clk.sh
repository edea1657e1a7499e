#!/bin/bash
# sample the shader clock / power while bench.py runs a long timed region
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
OUT=gpurun_out/clk_${1:-a}.log; : > $OUT
(for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|mclk|Power" | tr '\n' ' ' ; echo; sleep 0.25; done) > gpurun_out/clk_samples.txt &
SM=$!
DZ_NO_PROF=1 timeout 200 python bench.py --steps 3000 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-host-pass 2>&1 >/dev/null | grep "timed region" | cut -c1-200 >> $OUT
kill $SM 2>/dev/null
sort gpurun_out/clk_samples.txt | uniq -c | sort -rn | head -12 >> $OUT
cat $OUT
