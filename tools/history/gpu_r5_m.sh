#!/bin/bash
# round 5: the context table for k = 19 .. 25 (parity on large and repeat-rich graphs, rates at chr20 class against the generic kernel),
# and the even-k pass with its non-base walk on four loads in flight
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5m
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_graph_grid_variant or repeat_rich or even or other_odd or saturation or dense_hits" > gpurun_out/r5m/pytest.log 2>&1
tail -n 4 gpurun_out/r5m/pytest.log | cut -c1-200
python tools/bench_k.py --ks 27,20,22,24 > gpurun_out/r5m/bench_k.jsonl 2> gpurun_out/r5m/bench_k.err
cat gpurun_out/r5m/bench_k.jsonl | cut -c1-200; tail -2 gpurun_out/r5m/bench_k.err
for k in 27 25 23 21 19; do
  python tools/bench_large.py --k $k --check 1000000 2> gpurun_out/r5m/large_$k.err | tee gpurun_out/r5m/large_$k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:d[k] for k in ('k','n_keys','kernel_ms','reads_per_s','oracle_match','context_table')})"
done
for k in 25 21; do
  VGMI_CTABLE_K=0 python tools/bench_large.py --k $k --check 1000000 2> gpurun_out/r5m/generic_$k.err | tee gpurun_out/r5m/generic_$k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('generic', {k:d[k] for k in ('k','n_keys','kernel_ms','reads_per_s','oracle_match')})"
done
