#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
for seg in 32 48 64; do
  echo "stretch $seg KB"
  VGMI_GZ_SEG_KB=$seg timeout 600 python3 tools/bench_gzip_only.py 8000000 4 2,4 2>$O/e.log | grep '"device_gunzip": "1"' | cut -c1-140
done
