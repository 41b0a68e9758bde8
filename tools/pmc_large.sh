# counters of the large-graph (global filter) kernel on the chr20-class synthetic graph
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_large
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 2"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/p1 -o r1 -- python3 $ARGS > $OUT/b1.json 2> $OUT/e1.log
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum -d $OUT/p2 -o r1 -- python3 $ARGS > $OUT/b2.json 2> $OUT/e2.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/p3 -o r1 -- python3 $ARGS > $OUT/b3.json 2> $OUT/e3.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/p4 -o r1 -- python3 $ARGS > $OUT/b4.json 2> $OUT/e4.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep "count27" $OUT/summary.txt | awk '{print $1, $3, $4}'
tail -1 $OUT/b1.json | cut -c150-300
