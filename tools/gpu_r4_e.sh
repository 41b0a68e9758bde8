#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large_graph or repeat_rich" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
tail -3 $O/parity.log
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3"
C5="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2"
for w in 6 7 8; do
  VGMI_CT_WGS=$w timeout 300 python3 $C3 --check 500000 2>/dev/null > $O/c3_$w.json
  python3 -c "import sys,json; d=json.loads(open('$O/c3_$w.json').readline()); print('C3 wgs $w', d['kernel_ms'], d['all_kernel_ms'], d.get('oracle_match'), d['table_upload_s'])"
done
timeout 300 python3 tools/bench_large.py --genome 1200000000 --variants 2000000 --reads 40000000 --steps 3 --check 300000 2>/dev/null > $O/g12.json
python3 -c "import sys,json; d=json.loads(open('$O/g12.json').readline()); print('1.2Gb', d['kernel_ms'], d['all_kernel_ms'], d.get('oracle_match'), d['context_table'])"
for w in 6 8; do
  VGMI_CT_WGS=$w timeout 600 python3 $C5 2>/dev/null > $O/c5_$w.json
  python3 -c "import sys,json; d=json.loads(open('$O/c5_$w.json').readline()); print('C5 wgs $w', d['kernel_ms'], d['all_kernel_ms'], d['table_upload_s'])"
done
