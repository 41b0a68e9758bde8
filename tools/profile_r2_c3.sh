# C3 (chr20-class, table in HBM): kernel trace + memory-side counters of the large-table count kernel (count27x_kernel;
# VGMI_XTABLE=0 in the environment profiles count27_kernel<false, true> instead), separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; a calibration pass with the candidate runs dropped, VGMI_DBG=1,
# measures how FETCH_SIZE tallies this kernel's row stream)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_r2_c3
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r2 -- python3 $ARGS > $OUT/b0.json 2> $OUT/e0.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_main_fetch -o r2 -- python3 $ARGS > $OUT/b1.json 2> $OUT/e1.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_main_write -o r2 -- python3 $ARGS > $OUT/b2.json 2> $OUT/e2.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum -d $OUT/pmc_main_l2 -o r2 -- python3 $ARGS > $OUT/b4.json 2> $OUT/e4.log
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/pmc_main_ea -o r2 -- python3 $ARGS > $OUT/b5.json 2> $OUT/e5.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc_main_sq -o r2 -- python3 $ARGS > $OUT/b6.json 2> $OUT/e6.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
cat $OUT/summary.txt | grep -v "^$" | head -60
tail -2 $OUT/e5.log
