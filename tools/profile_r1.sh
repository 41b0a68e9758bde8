#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun):
#   kernel-trace stats of the default bench command, then PMC passes (counters only, separate runs)
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_r1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace_err.log
i=0
for pm in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
          "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
          "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pm -d $OUT/pmc_$i -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_$i.json 2> $OUT/pmc_${i}_err.log
done
# calibration of FETCH_SIZE on this kernel's own streaming pattern: same launch with the
# candidate runs dropped (VGMI_DBG=1: no table probes) reads exactly the read block once
VGMI_DBG=1 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_cal -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_cal.json 2> $OUT/pmc_cal_err.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
rm -f $OUT/*/*.db $OUT/*/*/*.db 2>/dev/null
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
