#!/bin/bash
# round 5: process-to-process spread of the whole-genome-class count kernel (29.1-34.5 ms on one box), by how the context
# table's 20 GB are allocated: hipMalloc, or one physical allocation mapped at an aligned virtual address (VGMI_CT_VMM=<MiB>)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5_spread; mkdir -p $OUT
N=${N:-5}
for mode in malloc vmm2 vmm1024; do
  for i in $(seq 1 $N); do
    case $mode in malloc) unset VGMI_CT_VMM;; vmm2) export VGMI_CT_VMM=2;; vmm1024) export VGMI_CT_VMM=1024;; esac
    VGMI_VERBOSE=1 python tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 3 2> $OUT/${mode}_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print(json.dumps({'mode':'$mode','run':$i,'kernel_ms':d['kernel_ms'],'all':d['all_kernel_ms'],'moved':d['context_table'].get('moved_entries'),'buckets':d['context_table'].get('n_buckets')}))" | tee -a $OUT/spread.jsonl
    grep "context table:" $OUT/${mode}_$i.err | tail -1
  done
done
