#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -q > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -30 $O/t.log | cut -c1-250
