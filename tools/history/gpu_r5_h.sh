#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5h
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_hmm.py tests/test_gpu_integration.py -x -q -m gpu --durations=3 > gpurun_out/r5h/integ.log 2>&1
echo "integ rc=$?" >> gpurun_out/r5h/integ.log
tail -8 gpurun_out/r5h/integ.log | cut -c1-200
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "not c3_chr20 and not c4_eight and not c5_scaled and not more_than_128" --durations=3 > gpurun_out/r5h/configs.log 2>&1
echo "configs rc=$?" >> gpurun_out/r5h/configs.log
tail -6 gpurun_out/r5h/configs.log | cut -c1-200
bash tools/profile_r5_c4.sh > gpurun_out/r5h/c4.txt 2>&1
head -50 gpurun_out/r5h/c4.txt | cut -c1-200; grep "done in" gpurun_out/r5h/c4.txt
N=4 bash tools/c5_spread.sh
