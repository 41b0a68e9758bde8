#!/bin/bash
# round 6, closing build: a second campaign of drawn end-to-end cases (the general draw from another seed, k = 28 and k = 26 forced on the 1.5 Mb genome)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_fuzz2; rm -rf $OUT; mkdir -p $OUT
python3 tools/fuzz_cli_parity.py 70000 300 > $OUT/fuzz_general.txt 2>&1; tail -1 $OUT/fuzz_general.txt
python3 tools/fuzz_cli_parity.py 71000 40 --k 28 --genome 1500000 > $OUT/fuzz_k28_large.txt 2>&1; tail -1 $OUT/fuzz_k28_large.txt
python3 tools/fuzz_cli_parity.py 72000 40 --k 26 --genome 1500000 > $OUT/fuzz_k26_large.txt 2>&1; tail -1 $OUT/fuzz_k26_large.txt
grep -h "^!!" $OUT/fuzz_*.txt | head -20
sha256sum varigraph_amd/libvgmi.so
