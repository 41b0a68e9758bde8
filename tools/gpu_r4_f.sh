#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4f
mkdir -p $O
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3"
C5="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2"
timeout 300 python3 $C3 --check 500000 2>/dev/null > $O/c3.json
python3 -c "import sys,json; d=json.loads(open('$O/c3.json').readline()); print('C3', d['kernel_ms'], d['all_kernel_ms'], d.get('oracle_match'), d['table_upload_s'])"
timeout 600 python3 $C5 2>/dev/null > $O/c5.json
python3 -c "import sys,json; d=json.loads(open('$O/c5.json').readline()); print('C5', d['kernel_ms'], d['all_kernel_ms'], d['table_upload_s'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_dist.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
