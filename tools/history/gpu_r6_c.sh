#!/bin/bash
# round 6: (1) BASELINE configs[4] through the CLI -- eight tetraploid samples of 2.6e8 read pairs each over the 3 Gb / 5 M variant graph, block-gzip
# input, one GPU (the samples name the same two files: 8 x 31 GB of distinct reads do not fit the box's disk) -- and (2) a campaign of drawn
# end-to-end cases through both CLIs: 150 of the general draw, 60 with k = 28 forced, 30 of them on the 1.5 Mb genome (> 65 536 k-mers)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_c; rm -rf $OUT; mkdir -p $OUT
( time python3 tools/wgs_cli_e2e.py --genome 3000000000 --contigs 24 --variants 5000000 --pairs 260000000 --files --bgzf --samples 8 ) > $OUT/e2e_wgs_tetraploid_8samples.json 2> $OUT/e2e.err
tail -3 $OUT/e2e.err | cut -c1-200; head -c 600 $OUT/e2e_wgs_tetraploid_8samples.json; echo
python3 tools/fuzz_cli_parity.py 60000 150 > $OUT/fuzz_general.txt 2>&1; tail -1 $OUT/fuzz_general.txt
python3 tools/fuzz_cli_parity.py 61000 30 --k 28 > $OUT/fuzz_k28.txt 2>&1; tail -1 $OUT/fuzz_k28.txt
python3 tools/fuzz_cli_parity.py 62000 30 --k 28 --genome 1500000 > $OUT/fuzz_k28_large.txt 2>&1; tail -1 $OUT/fuzz_k28_large.txt
grep -h "^!!" $OUT/fuzz_*.txt | head -20
