#!/bin/bash
# round 5, second sample: hipMalloc and VGMI_CT_VMM=2 processes in turn on one box (N of each), the whole-genome-class count kernel's ms
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5_spread2; mkdir -p $OUT
N=${N:-9}
for i in $(seq 1 $N); do
  for mode in malloc vmm2; do
    case $mode in malloc) unset VGMI_CT_VMM;; vmm2) export VGMI_CT_VMM=2;; esac
    VGMI_VERBOSE=1 python tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 3 2> $OUT/${mode}_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print(json.dumps({'mode':'$mode','run':$i,'kernel_ms':d['kernel_ms'],'all':d['all_kernel_ms']}))" | tee -a $OUT/spread.jsonl
  done
done
