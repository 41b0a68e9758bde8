#!/bin/bash
# (scratch) device tallies: CLI parity suites, then the c4 block with and without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4t; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_hmm.py tests/test_gpu_integration.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
for v in 1 0; do
  VGH_DEVICE_TALLIES=$v timeout 900 python bench.py --steps 20 --no-c3 --no-c5 --no-bloom --no-sample-level --no-cpu-baseline > $OUT/c4_$v.json 2> $OUT/c4_$v.err
  python3 -c "
import json
d=json.loads(open('$OUT/c4_$v.json').read().strip().splitlines()[-1])['c4']
print('tallies=$v', d['genotype_wall_s'], d['host_thread_seconds_per_sample'], d['host_thread_seconds'])
"
done
