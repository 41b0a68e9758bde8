#!/bin/bash
# Round-3 profiling recipe for the bench's dominant kernel (C2: count27s_kernel<true>), run on the GPU box through gpurun:
#   kernel-trace stats of the bench command, then PMC passes (counters only, one --pmc set per run), then a calibration pass of
#   FETCH_SIZE on the kernel's own row stream (ablation build, VGMI_DBG=1: candidate runs dropped -> the launch reads exactly the
#   read block once plus the per-workgroup staging of the filter)
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_r3
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-sample-level --no-c3 --no-c5 --verify-reads 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r3 -- python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-sample-level --c3-steps 10 --c5-steps 5 > $OUT/bench_trace.json 2> $OUT/trace_err.log
i=0
for pm in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
          "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA" \
          "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
          "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pm -d $OUT/pmc_$i -o r3 -- $B > $OUT/bench_pmc_$i.json 2> $OUT/pmc_${i}_err.log
done
VGMI_ABLATION=1 python3 -m varigraph_amd.build --force > /dev/null 2>&1
VGMI_DBG=1 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_cal -o r3 -- $B > $OUT/bench_pmc_cal.json 2> $OUT/pmc_cal_err.log
python3 -m varigraph_amd.build --force > /dev/null 2>&1
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep -v "^$" $OUT/summary.txt | grep -v "rows_kernel\|fq_\|inflate" | head -70
