#!/bin/bash
# round 5: even k = 20 .. 24 on graphs of more than 65 536 k-mers (context table + the debit pass): parity, rates at chr20 class
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_hmm.py -q -m gpu > gpurun_out/r5o/pytest.log 2>&1
tail -n 4 gpurun_out/r5o/pytest.log | cut -c1-200
