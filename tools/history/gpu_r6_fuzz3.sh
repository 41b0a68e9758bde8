#!/bin/bash
# round 6, closing build: a third, larger campaign of drawn end-to-end cases (800 of the general draw from seed 80000)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_fuzz3; rm -rf $OUT; mkdir -p $OUT
python3 tools/fuzz_cli_parity.py 80000 800 > $OUT/fuzz_general.txt 2>&1; tail -1 $OUT/fuzz_general.txt
grep -h "^!!" $OUT/fuzz_*.txt | head -20
sha256sum varigraph_amd/libvgmi.so
