#!/bin/bash
# round 4's closing validation: the whole GPU suite, smoke(), and the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4p
timeout 3300 python -m pytest tests -m gpu -x -q > gpurun_out/r4p/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r4p/gpu_suite.log
tail -15 gpurun_out/r4p/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/r4p/bench_n1.json 2> gpurun_out/r4p/bench_n1.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r4p/bench_n1.json
