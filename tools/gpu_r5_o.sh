#!/bin/bash
# round 5: even k = 20 .. 24 on graphs of more than 65 536 k-mers (context table + the debit pass): parity, rates at chr20 class
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5o
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_graph_grid_variant or repeat_rich or even or other_odd or saturation or dense_hits" > gpurun_out/r5o/pytest.log 2>&1
tail -n 4 gpurun_out/r5o/pytest.log | cut -c1-200
for k in 24 22 20; do
  python tools/bench_large.py --k $k --check 1000000 2> gpurun_out/r5o/large_$k.err | tee gpurun_out/r5o/large_$k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:d[k] for k in ('k','n_keys','kernel_ms','reads_per_s','oracle_match','context_table')})"
done
VGMI_CTABLE_K=0 python tools/bench_large.py --k 22 --reads 4000000 --check 1000000 2> gpurun_out/r5o/generic_22.err | tee gpurun_out/r5o/generic_22.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('literal', {k:d[k] for k in ('k','n_keys','kernel_ms','reads_per_s','oracle_match')})"
