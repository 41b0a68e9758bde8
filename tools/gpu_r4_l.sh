#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4l
timeout 2000 python3 tools/wgs_cli_e2e.py --genome 3000000000 --contigs 24 --variants 5000000 --pairs 100000000 --threads 16 > gpurun_out/r4l/wgs.json 2> gpurun_out/r4l/wgs.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4l/wgs.json')); lg=d.pop('genotype_log',[]); print(d.pop('construct_log',None)); print(d); print('\n'.join(lg[-30:]))"; tail -5 gpurun_out/r4l/wgs.err
