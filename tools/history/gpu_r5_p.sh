#!/bin/bash
# round 5: drawn end-to-end cases through both CLIs (graphs of more than 65 536 k-mers among them: the context table at k = 19 .. 27)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5p
timeout ${LIMIT:-2000} python3 tools/fuzz_cli_parity.py ${SEED:-6000} ${CASES:-70} > gpurun_out/r5p/fuzz_${SEED:-6000}.log 2>&1
grep -c "^ok" gpurun_out/r5p/fuzz_${SEED:-6000}.log; grep "^!!" gpurun_out/r5p/fuzz_${SEED:-6000}.log | cut -c1-700; tail -1 gpurun_out/r5p/fuzz_${SEED:-6000}.log | cut -c1-300
