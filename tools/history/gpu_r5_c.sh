#!/bin/bash
# round 5: even k = 20 .. 26 on the fast path (debit pass + grid kernel + exact tail); then the whole GPU suite with durations
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5c
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "other_odd_k or even_k or small_graph or cohort_counts or sample_pipeline" --durations=10 > gpurun_out/r5c/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/r5c/parity.log
tail -25 gpurun_out/r5c/parity.log
python tools/bench_k.py --ks 19,21,23,25,27,20,22,24,26,28 > gpurun_out/r5c/bench_k.jsonl 2> gpurun_out/r5c/bench_k.err
cat gpurun_out/r5c/bench_k.jsonl; tail -3 gpurun_out/r5c/bench_k.err
( time python -m pytest tests/ -x -q -m gpu --durations=40 ) > gpurun_out/r5c/suite.log 2>&1
tail -60 gpurun_out/r5c/suite.log
