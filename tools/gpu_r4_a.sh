#!/bin/bash
# round 4, first look at the context table: parity of the large-graph tests, then kernel ms at C3 / 1.2 Gb / C5 for both table forms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large_graph or repeat_rich" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
tail -5 $O/parity.log
for form in 1 0; do
  VGMI_CTABLE=$form timeout 600 python tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3 --check 1000000 > $O/c3_ct$form.json 2> $O/c3_ct$form.err
  tail -1 $O/c3_ct$form.json
done
VGMI_CTABLE=1 timeout 900 python tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2 > $O/c5_ct1.json 2> $O/c5_ct1.err
tail -1 $O/c5_ct1.json; tail -3 $O/c5_ct1.err
