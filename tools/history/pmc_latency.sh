cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/lat
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TCC|SQ)_[A-Z0-9_a-z]*(LATENCY|LAT|WAIT|STALL)[A-Za-z0-9_]*" | sort -u | head -80 > $OUT/names.txt
cat $OUT/names.txt | tr '\n' ' '
echo
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum -d $OUT/p1 -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b1.json 2> $OUT/e1.log
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_EXP_GDS -d $OUT/p2 -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b2.json 2> $OUT/e2.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep count27 $OUT/summary.txt | awk '{print $3, $4}'
grep -i "error\|invalid\|not found" $OUT/e1.log $OUT/e2.log | head -5
