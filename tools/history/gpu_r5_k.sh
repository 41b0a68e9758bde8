#!/bin/bash
# round 5: the scan form of the even-k debit pass (parity, rates), then the second sample of the whole-genome-class kernel's spread
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5k
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "even or other_odd or saturation" > gpurun_out/r5k/pytest.log 2>&1
tail -n 3 gpurun_out/r5k/pytest.log | cut -c1-200
python tools/bench_k.py --ks 27,20,22,24,21 > gpurun_out/r5k/bench_k.jsonl 2> gpurun_out/r5k/bench_k.err
cat gpurun_out/r5k/bench_k.jsonl | cut -c1-400; tail -3 gpurun_out/r5k/bench_k.err
N=${N:-8} bash tools/c5_spread2.sh
