#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4k
timeout 300 python -m pytest tests/test_gpu_ingest.py -q -x -k "named_pipe" 2>&1 | tail -3
timeout 600 python3 tools/wgs_cli_e2e.py --genome 30000000 --contigs 3 --variants 100000 --pairs 2000000 > gpurun_out/r4k/small.json 2> gpurun_out/r4k/small.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4k/small.json')); d.pop('genotype_log',None); d.pop('construct_log',None); print(d)"; tail -3 gpurun_out/r4k/small.err
timeout 900 python3 tools/wgs_cli_e2e.py --genome 300000000 --contigs 24 --variants 500000 --pairs 10000000 > gpurun_out/r4k/mid.json 2> gpurun_out/r4k/mid.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4k/mid.json')); lg=d.pop('genotype_log',[]); d.pop('construct_log',None); print(d); print('\n'.join(lg[-25:]))"; tail -3 gpurun_out/r4k/mid.err
