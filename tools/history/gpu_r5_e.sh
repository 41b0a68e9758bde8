#!/bin/bash
# round 5: flagged rows scored again on the device (sequence fixes), eight consumers; the suites that hold VCFs against the reference; the c4 stage profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5e
python tools/bench_k.py --ks 27,22 > gpurun_out/r5e/bench_k.jsonl 2> gpurun_out/r5e/bench_k.err
cat gpurun_out/r5e/bench_k.jsonl
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_integration.py tests/test_gpu_hmm.py -x -q -m gpu --durations=8 > gpurun_out/r5e/integ.log 2>&1
echo "integ rc=$?" >> gpurun_out/r5e/integ.log
tail -14 gpurun_out/r5e/integ.log | cut -c1-200
python -m pytest tests/test_gpu_configs.py -x -q -m gpu --durations=8 > gpurun_out/r5e/configs.log 2>&1
echo "configs rc=$?" >> gpurun_out/r5e/configs.log
tail -14 gpurun_out/r5e/configs.log | cut -c1-200
bash tools/profile_r5_c4.sh | head -60 | cut -c1-330
