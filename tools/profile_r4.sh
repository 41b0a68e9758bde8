#!/bin/bash
# Round-4 profiling recipe for the large-graph count kernel (count27c_kernel over the context table), C3 and C5 in SEPARATE runs so
# that every roofline figure of the bench line's c3 / c5 blocks can be recomputed from profiles/: kernel-trace stats, then PMC passes
# (counters only, one --pmc set per run).  Usage: tools/profile_r4.sh c3|c5
set -u
W=${1:-c3}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_r4_$W
rm -rf $OUT; mkdir -p $OUT
if [ $W = c3 ]; then A="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3"
else A="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 1"; fi
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r4 -- python3 $A > $OUT/b0.json 2> $OUT/e0.log
i=0
for pm in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
          "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-include-regex "count27" --pmc $pm -d $OUT/pmc_$i -o r4 -- python3 $A > $OUT/b$i.json 2> $OUT/e$i.log
done
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep -v "^$" $OUT/summary.txt | grep "count27\|PMC\|kernel-trace\|calls" | head -60
