#!/bin/bash
# End-of-round visit: full GPU test suite, smoke(), a two-rank rehearsal of bench.py on the one GPU
# (gloo), isolated kernel timings, then the profiling visit.   usage: tools/gpu_final.sh <tag>
TAG=${1:-r02_b}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== pytest -m gpu"
timeout -s KILL 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
echo "=== smoke"
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "=== 2 ranks on one GPU (gloo)"
DZ_DIST_BACKEND=gloo DZ_FORCE_DEVICE=0 timeout -s KILL 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline --no-exact-f32 \
    2> gpurun_out/two_rank_$TAG.err | cut -c1-260
tail -2 gpurun_out/two_rank_$TAG.err | cut -c1-200
echo "=== kbench"
timeout -s KILL 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids | grep " us " | cut -c1-70
cp gpurun_out/kbench.json gpurun_out/kbench_$TAG.json
echo "=== profile visit"
bash tools/gpu_bench.sh $TAG 200 2>&1 | tail -22 | cut -c1-260
