#!/bin/bash
# Round 4, K3 in its binned form: kernel table and HBM traffic of one 60 Mb call (separate --pmc passes), the direct form beside
# it, and the whole-genome-class filter (3 Gb, 28.8 GB of counters) once, added in chromosome-sized pieces.
# Results: gpurun_out/r4_bloom/ (summary.txt -> profiles/r4_bloom_rocprofv3_summary.txt, bloom.jsonl -> profiles/r4_bloom.jsonl)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4_bloom; rm -rf $OUT; mkdir -p $OUT
: > $OUT/bloom.jsonl
for b in 1 0; do VGMI_BLOOM_BINNED=$b python3 tools/bench_bloom.py --genome 60000000 --steps 3 2> /dev/null | sed "s/^{/{\"binned\": $b, /" >> $OUT/bloom.jsonl; done
for b in 1 0; do VGMI_BLOOM_BINNED=$b timeout 900 python3 tools/bench_bloom.py --genome 3000000000 --piece 250000000 --steps 1 2> $OUT/wgs_b$b.err | sed "s/^{/{\"binned\": $b, /" >> $OUT/bloom.jsonl; done
cat $OUT/bloom.jsonl
ARGS="tools/bench_bloom.py --genome 60000000 --steps 2"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 $ARGS > $OUT/kt.json 2> $OUT/kt.log
rocprofv3 --kernel-include-regex "bb_|rows_kernel" --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r -- python3 $ARGS > $OUT/f.json 2> $OUT/f.log
rocprofv3 --kernel-include-regex "bb_|rows_kernel" --pmc WRITE_SIZE -d $OUT/pmc_write -o r -- python3 $ARGS > $OUT/w.json 2> $OUT/w.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep -i "bb_\|rows_kernel" $OUT/summary.txt | cut -c1-150
