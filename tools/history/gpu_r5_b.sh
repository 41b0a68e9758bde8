#!/bin/bash
# round 5: small graphs of odd k = 19 .. 25 on the path-table kernel (grid of 8): parity, then the rates of tools/bench_k.py
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "other_odd_k or small_graph or every_kmer_counted or table_lookup or sketch_keys" --durations=10 > gpurun_out/r5b/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/r5b/parity.log
tail -15 gpurun_out/r5b/parity.log
python -m pytest tests/test_gpu_gunzip.py -x -q -m gpu > gpurun_out/r5b/gunzip.log 2>&1
tail -3 gpurun_out/r5b/gunzip.log
python tools/bench_k.py --ks 19,21,23,25,27,22,28 > gpurun_out/r5b/bench_k.jsonl 2> gpurun_out/r5b/bench_k.err
cat gpurun_out/r5b/bench_k.jsonl; tail -3 gpurun_out/r5b/bench_k.err
VGMI_SMALLK=0 python tools/bench_k.py --ks 21,25 > gpurun_out/r5b/bench_k_generic.jsonl 2>> gpurun_out/r5b/bench_k.err
cat gpurun_out/r5b/bench_k_generic.jsonl
