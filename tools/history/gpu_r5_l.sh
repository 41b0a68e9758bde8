#!/bin/bash
# round 5: kernel trace of the k sweep (the even-k pass in its scan form), then drawn end-to-end cases through both CLIs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash tools/profile_r5.sh k > gpurun_out/r5l_profile.log 2>&1
cp gpurun_out/prof_r5_k/summary.txt gpurun_out/r5l_k_summary.txt
head -12 gpurun_out/r5l_k_summary.txt | cut -c1-160
mkdir -p gpurun_out/r5l
timeout 1500 python3 tools/fuzz_cli_parity.py ${SEED:-5000} ${CASES:-60} > gpurun_out/r5l/fuzz.log 2>&1
grep -c "^ok" gpurun_out/r5l/fuzz.log; grep "^!!" gpurun_out/r5l/fuzz.log | cut -c1-600; tail -2 gpurun_out/r5l/fuzz.log | cut -c1-300
