#!/bin/bash
# round 6, the closing build: the GPU suite as the driver runs it, the default bench line, the counter profiles of the count passes and of K3
# (stamped with this library's sha256: bench.py reports `traffic` from them), rates by k on a small and a chr20-class graph
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_final; rm -rf $OUT; mkdir -p $OUT
( time python -m pytest tests/ -x -q -m gpu ) > $OUT/suite.log 2>&1
tail -n 4 $OUT/suite.log | cut -c1-200
TAG=r6 bash tools/profile_r6.sh c3 c5 c2 bloom > $OUT/profile.log 2>&1
grep "GB per launch" $OUT/profile.log
for w in c3 c5 c2 bloom; do cp gpurun_out/prof_r6_$w/traffic.json $OUT/traffic_$w.json; cp gpurun_out/prof_r6_$w/summary.txt $OUT/summary_$w.txt; done
# the bench line WITH the profiles of this build in place (profiles/hbm_traffic*.json are what the line reads)
cp $OUT/traffic_c3.json profiles/hbm_traffic_c3.json; cp $OUT/traffic_c5.json profiles/hbm_traffic_c5.json; cp $OUT/traffic_c2.json profiles/hbm_traffic.json; cp $OUT/traffic_bloom.json profiles/hbm_traffic_bloom.json
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -n 3 $OUT/bench.err | cut -c1-200
python3 tools/bench_k.py --ks 19,21,23,25,27,20,22,24,26,28 > $OUT/bench_k.jsonl 2> $OUT/bench_k.err; cut -c1-170 $OUT/bench_k.jsonl
for k in 27 25 23 21 19 28 26 24 22 20; do python3 tools/bench_large.py --k $k --steps 3 --check 500000 2>> $OUT/err.log | tail -1 >> $OUT/bench_large_k.jsonl; done
python3 -c "
import json
for ln in open('$OUT/bench_large_k.jsonl'):
    d = json.loads(ln); print('k', d['k'], 'kernel_ms', round(d['kernel_ms'], 3), 'reads/s', round(d['reads_per_s'] / 1e9, 3), 'oracle', d.get('oracle_match'))"
