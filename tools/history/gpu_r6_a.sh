#!/bin/bash
# round 6, A/B of the deferred counter updates (vgmi_ctdefer.hip) on one box: parity cases, then the chr20-class launch with the
# counts in the row loop and deferred, in turn, then the deferred form's kernel table
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_a; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deferred or counts-in-the-row-loop" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for i in 1 2 3; do
  for d in 0 1; do
    VGMI_CT_DEFER=$d python3 tools/bench_large.py --steps 4 $( [ $i = 1 ] && echo --check 2000000 ) 2>> $OUT/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$d', 'kernel_ms', [round(x, 3) for x in d['all_kernel_ms']], 'oracle', d.get('oracle_match'))" | tee -a $OUT/ab.txt
  done
done
VGMI_CT_DEFER=1 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 tools/bench_large.py --steps 3 > $OUT/traced.json 2>> $OUT/err.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
head -12 $OUT/summary.txt | cut -c1-160
