# usage: bash tools/pmc_quick.sh [VGMI_DBG value]   -- two counter passes of the default bench command
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export VGMI_DBG=${1:-0}
OUT=gpurun_out/pmcq_$VGMI_DBG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH -d $OUT/p1 -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b1.json 2> $OUT/e1.log
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_BUSY_CYCLES -d $OUT/p2 -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b2.json 2> $OUT/e2.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
echo "VGMI_DBG=$VGMI_DBG"; grep count27 $OUT/summary.txt | awk '{print $3, $4}'
tail -2 $OUT/e2.log | cut -c1-200
