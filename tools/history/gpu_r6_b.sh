#!/bin/bash
# round 6: k = 28 through the context table (parity cases, rates on a small and a chr20-class graph), the two-rank bench with its c3 / c5
# legs, the fix_j test, then the default bench line of the rewritten bench.py
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_b; rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_hmm.py -x -q -m gpu -k "28 or bench_starts or beyond_65535" > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log | cut -c1-200
python3 tools/bench_k.py --ks 27,26,28 > $OUT/bench_k.jsonl 2> $OUT/bench_k.err; cut -c1-200 $OUT/bench_k.jsonl
for k in 27 26 28; do python3 tools/bench_large.py --k $k --steps 3 --check 1000000 2>> $OUT/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k', d['k'], 'kernel_ms', round(d['kernel_ms'], 3), 'reads/s', round(d['reads_per_s'] / 1e9, 3), 'oracle', d.get('oracle_match'), d['context_table'])" | tee -a $OUT/large_k.txt; done
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -n 3 $OUT/bench.err | cut -c1-300
head -c 1500 $OUT/bench.json
