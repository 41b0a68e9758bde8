# usage: bash tools/pmc_c2.sh <tag>   -- instruction-mix counters of the C2 count kernel (env such as VGMI_GRID12 / VGMI_PTABLE / VGMI_DBG is inherited)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_c2_$1
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-c3 --no-c5 --no-sample-level --verify-reads 0"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH -d $OUT/p1 -o r1 -- $B > $OUT/b1.json 2> $OUT/e1.log
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU -d $OUT/p2 -o r1 -- $B > $OUT/b2.json 2> $OUT/e2.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep "count27" $OUT/summary.txt | awk '{printf "%-22s %14.0f\n", $4, $3}'
