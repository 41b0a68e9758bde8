#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
for mb in 256 512 1024; do
  echo "text chunk $mb MB"
  VGMI_FASTQ_TEXT_MB=$mb timeout 600 python3 tools/bench_gzip_only.py 8000000 4 4 2>$O/e.log | grep '"device_gunzip": "1"' | cut -c1-140
  VGMI_FASTQ_TEXT_MB=$mb timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 512 2>>$O/e.log | cut -c1-140
done
