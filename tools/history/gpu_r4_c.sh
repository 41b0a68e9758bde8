#!/bin/bash
# round 4: context table with the deferred trail (pending ring) -- parity of the large-graph tests, then kernel ms by load factor
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large_graph or repeat_rich" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
tail -3 $O/parity.log
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3"
for load in 25 30 40; do
  VGMI_CTABLE_LOAD=$load timeout 300 python3 $C3 --check 500000 2>/dev/null > $O/c3_load$load.json
  python3 -c "import sys,json; d=json.loads(open('$O/c3_load$load.json').readline()); print('C3 load $load', d['kernel_ms'], d['all_kernel_ms'], d['context_table'], d.get('oracle_match'), d['table_upload_s'])"
done
for load in 30 40; do
  VGMI_CTABLE_LOAD=$load timeout 600 python3 tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2 2>/dev/null > $O/c5_load$load.json
  python3 -c "import sys,json; d=json.loads(open('$O/c5_load$load.json').readline()); print('C5 load $load', d['kernel_ms'], d['all_kernel_ms'], d['context_table'], d['table_upload_s'])"
done
