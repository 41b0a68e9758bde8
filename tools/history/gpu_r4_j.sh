#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4j
bash tools/profile_r4.sh c3 > gpurun_out/r4j/prof_c3.log 2>&1
tail -30 gpurun_out/r4j/prof_c3.log
bash tools/profile_r4.sh c5 > gpurun_out/r4j/prof_c5.log 2>&1
tail -30 gpurun_out/r4j/prof_c5.log
timeout 900 python3 bench.py > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err
tail -c 3000 gpurun_out/r4j/bench.json
tail -5 gpurun_out/r4j/bench.err
