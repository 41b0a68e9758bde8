#!/bin/bash
# round 5: even k with the debit pass straight from global memory; the c4 stage profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "other_odd_k or even_k" --durations=5 > gpurun_out/r5d/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/r5d/parity.log
tail -12 gpurun_out/r5d/parity.log | cut -c1-250
python tools/bench_k.py --ks 27,20,22,24,26 > gpurun_out/r5d/bench_k.jsonl 2> gpurun_out/r5d/bench_k.err
cat gpurun_out/r5d/bench_k.jsonl; tail -3 gpurun_out/r5d/bench_k.err
bash tools/profile_r5_c4.sh
