#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5g
bash tools/profile_r5_c4.sh > gpurun_out/r5g/c4.txt 2>&1
head -34 gpurun_out/r5g/c4.txt | cut -c1-200; grep "HMM part\|genotyping\|done in\|counting\|around the device" gpurun_out/r5g/c4.txt | tail -40 | cut -c1-430
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q bash tools/profile_r5_c4.sh > gpurun_out/r5g/c4_q$q.txt 2>&1
  echo "== GPU_MAX_HW_QUEUES=$q"; grep "genotype_wall_s\|counting_wall" gpurun_out/r5g/c4_q$q.txt | head -2
done
