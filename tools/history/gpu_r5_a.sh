#!/bin/bash
# round 5, first GPU call: ADVICE fixes (gzip scratch regrowth, trailer check, member loop), --procs restructure
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu --durations=10 > gpurun_out/r5a/gunzip.log 2>&1
echo "gunzip rc=$?" >> gpurun_out/r5a/gunzip.log
tail -5 gpurun_out/r5a/gunzip.log
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c4_eight" -s > gpurun_out/r5a/c4.log 2>&1
echo "c4 rc=$?" >> gpurun_out/r5a/c4.log
tail -15 gpurun_out/r5a/c4.log
python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r5a/dist.log 2>&1
tail -3 gpurun_out/r5a/dist.log
