# memory-side counters of the large-graph kernel; table placement knobs come from the environment (VGMI_LOCALITY, ...)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_large_${1:-x}
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 2"
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/p2 -o r1 -- python3 $ARGS > $OUT/b2.json 2> $OUT/e2.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/p3 -o r1 -- python3 $ARGS > $OUT/b3.json 2> $OUT/e3.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/p4 -o r1 -- python3 $ARGS > $OUT/b4.json 2> $OUT/e4.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep "count27" $OUT/summary.txt | awk '{print $1, $3, $4}'
tail -1 $OUT/b2.json | cut -c150-300
