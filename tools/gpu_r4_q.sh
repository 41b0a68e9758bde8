#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4q
timeout 900 python -m pytest tests/test_gpu_gunzip.py -x -q > gpurun_out/r4q/t.log 2>&1; echo "rc=$?" >> gpurun_out/r4q/t.log
tail -40 gpurun_out/r4q/t.log
