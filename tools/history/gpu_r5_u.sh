#!/bin/bash
# round 5: the whole-genome-class graph (3 Gb, 5 M SNPs) at k = 21 and k = 22 through the context table: rates (no oracle leg at this size here;
# the chr20-class runs and the suite carry the parity)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5u
for k in 21 22; do
  python tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2 --k $k 2> gpurun_out/r5u/wgs_$k.err | tee gpurun_out/r5u/wgs_$k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:d[k] for k in ('k','n_keys','kernel_ms','reads_per_s','context_table','table_upload_s')})"
done
