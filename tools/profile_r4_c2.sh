#!/bin/bash
# Round 4: kernel table of the default bench command's own leg (C2: count27s_kernel<true>, unchanged since round 3), so that the
# bench line's roofline.achieved can be recomputed from profiles/ (kernel average x launches).  -> profiles/r4_rocprofv3_summary.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4_c2; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 bench.py --no-c3 --no-c5 --no-bloom --no-sample-level --no-c4 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
head -12 $OUT/summary.txt | cut -c1-150; python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['value'])"
