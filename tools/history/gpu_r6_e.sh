#!/bin/bash
# round 6: two partition levels for the deferred counter updates of whole-genome-class tables: parity cases, the C5 tests of the suite, then the
# whole-genome-class launch with the counter updates in the row loop and deferred in turn (+ the kernel table of the deferred form)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_e; rm -rf $OUT; mkdir -p $OUT
# (needs tools/history/ctd_two_levels_experiment.patch applied: the experiment is not in the tree)
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -x -q -m gpu -k "deferred or wgs or c5 or c3_full" > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log | cut -c1-160
A="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000"
for i in 1 2; do
  for d in 0 1; do
    VGMI_CT_DEFER=$d python3 $A --steps 3 $( [ $i = 1 ] && echo --check 100000 ) 2>> $OUT/err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$d', 'kernel_ms', [round(x, 3) for x in d['all_kernel_ms']], 'oracle', d.get('oracle_match'))" | tee -a $OUT/ab.txt
  done
done
VGMI_CT_DEFER=1 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 $A --steps 2 > $OUT/traced.json 2>> $OUT/err.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep "countkc\|count27c\|ctd_" $OUT/summary.txt | cut -c1-150
