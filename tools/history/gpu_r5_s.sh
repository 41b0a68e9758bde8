#!/bin/bash
# round 5, closing: kernel tables + traffic counters of the context table at other k, then the whole suite once more
cd "$(dirname "$0")/.."
bash tools/profile_r5.sh largek > gpurun_out/r5s_profile.log 2>&1
tail -40 gpurun_out/r5s_profile.log | cut -c1-170
bash tools/gpu_r5_suite.sh
