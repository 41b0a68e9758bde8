#!/bin/bash
# round 5: the even-k pass with its list of non-bases and walk launch: parity, kernel trace, rates
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r5n
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "even or other_odd or saturation" > gpurun_out/r5n/pytest.log 2>&1
tail -n 3 gpurun_out/r5n/pytest.log | cut -c1-200
OUT=gpurun_out/prof_r5_k2; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 tools/bench_k.py --ks 27,22 > $OUT/bench_k_traced.jsonl 2> $OUT/err.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
head -12 $OUT/summary.txt | cut -c1-150
python tools/bench_k.py --ks 19,21,23,25,27,20,22,24,26,28 > gpurun_out/r5n/bench_k.jsonl 2> gpurun_out/r5n/bench_k.err
cut -c1-170 gpurun_out/r5n/bench_k.jsonl
