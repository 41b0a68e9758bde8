# SQ counters of the device inflate kernel (separate passes), block-gzip sample path
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_bgzf; rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_bgzf_only.py 4000000 4 100"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_a -o r -- python3 $ARGS > $OUT/a.json 2> $OUT/a.log
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH -d $OUT/pmc_b -o r -- python3 $ARGS > $OUT/b.json 2> $OUT/b.log
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC -d $OUT/pmc_c -o r -- python3 $ARGS > $OUT/c.json 2> $OUT/c.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep -i -A14 "bgzf_inflate" $OUT/summary.txt | head -70; tail -2 $OUT/c.log
