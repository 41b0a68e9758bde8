#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4o
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -x -s -k "c4_eight or c5_scaled" > gpurun_out/r4o/t.log 2>&1; echo "rc=$?" >> gpurun_out/r4o/t.log
grep -v "^$" gpurun_out/r4o/t.log | tail -25
