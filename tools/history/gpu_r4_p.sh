#!/bin/bash
# round 4's closing validation: the whole GPU suite, smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4p
timeout 3300 python -m pytest tests -m gpu -x -q > gpurun_out/r4p/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r4p/gpu_suite.log
tail -8 gpurun_out/r4p/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/r4p/bench_n1.json 2> gpurun_out/r4p/bench_n1.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r4p/bench_n1.json').read().strip().splitlines()[-1])
print('C2', d['roofline']['frac'], 'c3', d['c3']['roofline']['frac'], 'c5', d['c5']['roofline']['frac'], 'bloom', d['bloom']['add_kmers_per_s'])
print({k:v for k,v in d['sample_level'].items() if 'reads_per_s' in k})
"
