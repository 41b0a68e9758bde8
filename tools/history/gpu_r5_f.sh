#!/bin/bash
# round 5: the shared part cache and the device-resident plan; tests that hold VCFs against the reference; the c4 stage profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5f
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_hmm.py tests/test_gpu_integration.py -x -q -m gpu --durations=5 > gpurun_out/r5f/integ.log 2>&1
echo "integ rc=$?" >> gpurun_out/r5f/integ.log
tail -12 gpurun_out/r5f/integ.log | cut -c1-200
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "not c3_chr20 and not c4_eight and not c5_scaled and not more_than_128" --durations=5 > gpurun_out/r5f/configs.log 2>&1
echo "configs rc=$?" >> gpurun_out/r5f/configs.log
tail -8 gpurun_out/r5f/configs.log | cut -c1-200
bash tools/profile_r5_c4.sh > gpurun_out/r5f/c4.txt 2>&1
head -45 gpurun_out/r5f/c4.txt | cut -c1-200; grep "HMM part\|genotyping\|done in\|counting" gpurun_out/r5f/c4.txt | tail -40 | cut -c1-420
