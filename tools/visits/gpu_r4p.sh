#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "--- packed"; timeout 300 python tools/rec_contention.py --out gpurun_out/rec_contention_pk.json 2>&1 | grep -v amdgpu | tail -1 | cut -c1-700
echo "--- plain"; DZ_LSTM_PK=0 timeout 300 python tools/rec_contention.py --out gpurun_out/rec_contention_plain.json 2>&1 | grep -v amdgpu | tail -1 | cut -c1-700
DZ_LSTM_PK=0 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "lstm" -p no:cacheprovider 2>&1 | tail -2
GRID="DZ_LSTM_PK=1 DZ_LSTM_PK=0 DZ_LSTM_PK=1 DZ_LSTM_PK=0 DZ_LSTM_PK=0,DZ_GEMM_GEN=2" bash tools/visits/gpu_r4n.sh
