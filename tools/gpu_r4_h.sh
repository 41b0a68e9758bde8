#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large_graph or repeat_rich" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
tail -3 $O/parity.log
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 4"
C5="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2"
for ct in 1 0 1; do
  VGMI_CTABLE=$ct timeout 300 python3 $C3 --check 500000 2>/dev/null > $O/c3.json
  python3 -c "import sys,json; d=json.loads(open('$O/c3.json').readline()); print('C3 ctable $ct:', round(d['kernel_ms'],3), [round(x,2) for x in d['all_kernel_ms']], d.get('oracle_match'))"
done
for ct in 1 0 1; do
  VGMI_CTABLE=$ct timeout 600 python3 $C5 2>/dev/null > $O/c5.json
  python3 -c "import sys,json; d=json.loads(open('$O/c5.json').readline()); print('C5 ctable $ct:', round(d['kernel_ms'],3), [round(x,2) for x in d['all_kernel_ms']])"
done
timeout 1200 python -m pytest tests/test_gpu_large.py tests/test_gpu_dist.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
