#!/bin/bash
# round 4: where count27c_kernel's time goes (ablation build: wrong counters on purpose) + memory-side counters, C3 and C5
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b
mkdir -p $O
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 2"
C5="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2"
VGMI_ABLATION=1 python3 -m varigraph_amd.build --force > /dev/null 2>&1
for d in 0 1 2 3 4 8 12 15; do
  echo "dbg=$d" >> $O/abl_c3.txt
  VGMI_DBG=$d timeout 300 python3 $C3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['kernel_ms'], d['all_kernel_ms'])" >> $O/abl_c3.txt
done
cat $O/abl_c3.txt
for d in 0 1 2 8 15; do
  echo "dbg=$d" >> $O/abl_c5.txt
  VGMI_DBG=$d timeout 600 python3 $C5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['kernel_ms'], d['all_kernel_ms'])" >> $O/abl_c5.txt
done
cat $O/abl_c5.txt
python3 -m varigraph_amd.build --force > /dev/null 2>&1
for W in c3 c5; do
  A=$C3; [ $W = c5 ] && A=$C5
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum -d $O/pmc_${W}_l2 -o r4 -- python3 $A > $O/${W}_b1.json 2> $O/${W}_e1.log
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_${W}_ea -o r4 -- python3 $A > $O/${W}_b2.json 2> $O/${W}_e2.log
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU -d $O/pmc_${W}_sq -o r4 -- python3 $A > $O/${W}_b3.json 2> $O/${W}_e3.log
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -d $O/pmc_${W}_tcp -o r4 -- python3 $A > $O/${W}_b4.json 2> $O/${W}_e4.log
done
python3 tools/rocprof_summary.py $O > $O/summary.txt
find $O -name "*.db" -delete
grep "count27" $O/summary.txt | head -60
tail -2 $O/c5_e4.log
