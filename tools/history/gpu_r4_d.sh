#!/bin/bash
# round 4: counters aligned to sectors (VGMI_CTABLE_ALIGN) and workgroups per CU (VGMI_CT_WGS), chr20 and whole-genome class
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large_graph or repeat_rich" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
tail -3 $O/parity.log
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3"
C5="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2"
for cfg in "16 8" "1 8" "16 7" "16 6" "16 5"; do
  set -- $cfg
  VGMI_CTABLE_ALIGN=$1 VGMI_CT_WGS=$2 timeout 300 python3 $C3 --check 500000 2>/dev/null > $O/c3_$1_$2.json
  python3 -c "import sys,json; d=json.loads(open('$O/c3_$1_$2.json').readline()); print('C3 align $1 wgs $2', d['kernel_ms'], d['all_kernel_ms'], d.get('oracle_match'), d['table_upload_s'])"
done
for cfg in "16 8" "1 8" "16 6"; do
  set -- $cfg
  VGMI_CTABLE_ALIGN=$1 VGMI_CT_WGS=$2 timeout 600 python3 $C5 2>/dev/null > $O/c5_$1_$2.json
  python3 -c "import sys,json; d=json.loads(open('$O/c5_$1_$2.json').readline()); print('C5 align $1 wgs $2', d['kernel_ms'], d['all_kernel_ms'], d['table_upload_s'])"
done
