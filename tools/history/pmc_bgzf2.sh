cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_bgzf2; rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_bgzf_only.py 4000000 4 100"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SALU SQ_WAVE_CYCLES -d $OUT/pmc_a -o r -- python3 $ARGS > $OUT/a.json 2> $OUT/a.log
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc_b -o r -- python3 $ARGS > $OUT/b.json 2> $OUT/b.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep "bgzf_inflate" $OUT/summary.txt | awk '{printf "%-24s total %-16s avg %s\n", $4, $2, $3}'; tail -1 $OUT/b.log
