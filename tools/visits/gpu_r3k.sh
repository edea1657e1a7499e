#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== kbench lstm"
for nc in 1 2; do DZ_LSTM_NC=$nc timeout 100 python tools/kbench.py --only lstm 2>&1 | grep "^lstm " | sed "s/^/NC=$nc /"; done
echo "=== tests (default NC)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_r2.py tests/test_gpu_pipeline.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
echo "=== tests (DZ_LSTM_NC=2 forced: small batches, odd batches)"
DZ_LSTM_NC=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_r2.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
SKIP_TESTS=1 NK=6 bash tools/gpu_ab.sh r3k none "DZ_LSTM_NC=2 DZ_LSTM_NC=1 DZ_LSTM_NC=2 DZ_LSTM_NC=1 DZ_LSTM_NC=2,DZ_DEPTH=3" | cut -c1-150
