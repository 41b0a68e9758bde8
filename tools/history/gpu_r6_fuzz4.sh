#!/bin/bash
# round 6, the closing library (8a23f251...): a fourth campaign of drawn end-to-end cases (600 of the general draw from seed 90000, 60 with k = 28 forced)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_fuzz4; rm -rf $OUT; mkdir -p $OUT
python3 tools/fuzz_cli_parity.py 90000 600 > $OUT/fuzz_general.txt 2>&1; tail -1 $OUT/fuzz_general.txt
python3 tools/fuzz_cli_parity.py 91000 60 --k 28 > $OUT/fuzz_k28.txt 2>&1; tail -1 $OUT/fuzz_k28.txt
grep -h "^!!" $OUT/fuzz_*.txt | head -20
sha256sum varigraph_amd/libvgmi.so
