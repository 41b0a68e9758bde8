#!/bin/bash
# round 5: k = 23 .. 26 on the context table with three lookups per pair of lanes instead of four: parity, rates at chr20 class
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5w
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_graph_grid_variant or repeat_rich or even or other_odd or saturation or dense_hits" > gpurun_out/r5w/pytest.log 2>&1
tail -n 3 gpurun_out/r5w/pytest.log | cut -c1-200
for k in 27 25 23 26 24 21; do
  python tools/bench_large.py --k $k --check 1000000 2> gpurun_out/r5w/large_$k.err | tee gpurun_out/r5w/large_$k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:d[k] for k in ('k','n_keys','kernel_ms','reads_per_s','oracle_match')})"
done
python tools/check_wgs_k.py 25,26 100000 2> gpurun_out/r5w/wgs_check.err | tee gpurun_out/r5w/wgs_check.jsonl | cut -c1-200
python tools/bench_k.py --ks 27,26 2> /dev/null | cut -c1-200
