#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5j
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_hmm.py tests/test_gpu_integration.py -x -q -m gpu --durations=3 > gpurun_out/r5j/integ.log 2>&1
echo "integ rc=$?" >> gpurun_out/r5j/integ.log; tail -6 gpurun_out/r5j/integ.log | cut -c1-200
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "not c3_chr20 and not c4_eight and not c5_scaled and not more_than_128" --durations=3 > gpurun_out/r5j/configs.log 2>&1
echo "configs rc=$?" >> gpurun_out/r5j/configs.log; tail -5 gpurun_out/r5j/configs.log | cut -c1-200
bash tools/profile_r5_c4.sh > gpurun_out/r5j/c4.txt 2>&1
head -48 gpurun_out/r5j/c4.txt | cut -c1-200; grep "done in" gpurun_out/r5j/c4.txt
bash tools/profile_r5.sh hmm
