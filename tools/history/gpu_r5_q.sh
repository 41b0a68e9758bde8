#!/bin/bash
# round 5: k = 26 through the context table at any size (parity, rates), then a campaign of drawn end-to-end cases
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5q
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_graph_grid_variant or repeat_rich or even or other_odd or saturation or dense_hits or sketch" > gpurun_out/r5q/pytest.log 2>&1
tail -n 3 gpurun_out/r5q/pytest.log | cut -c1-200
python tools/bench_k.py --ks 27,26,28 > gpurun_out/r5q/bench_k.jsonl 2> gpurun_out/r5q/bench_k.err
cut -c1-170 gpurun_out/r5q/bench_k.jsonl
python tools/bench_large.py --k 26 --check 1000000 2> gpurun_out/r5q/large_26.err | tee gpurun_out/r5q/large_26.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:d[k] for k in ('k','n_keys','kernel_ms','reads_per_s','oracle_match')})"
timeout ${LIMIT:-1500} python3 tools/fuzz_cli_parity.py ${SEED:-8000} ${CASES:-500} > gpurun_out/r5q/fuzz.log 2>&1
grep -c "^ok" gpurun_out/r5q/fuzz.log; grep "^!!" gpurun_out/r5q/fuzz.log | cut -c1-700; tail -1 gpurun_out/r5q/fuzz.log | cut -c1-300
