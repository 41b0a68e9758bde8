#!/bin/bash
# (ran against a working build with VGMI_CT_DIFF: counters as a difference array over the places; measured slower, 8.82 against 8.44 ms, the
# code is gone.  Kept as the record of how the same-box comparisons were run: every configuration three times, interleaved.)
# same-box A/B, interleaved, three rounds: difference counters vs plain, marks per X vs one, against round 2's table as the box's yardstick
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4i
mkdir -p $O
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 4"
for rep in 1 2 3; do
for cfg in "1 1 1" "1 0 1" "1 1 0" "0 1 1"; do
  set -- $cfg
  VGMI_CTABLE=$1 VGMI_CT_DIFF=$2 VGMI_CT_MARKS=$3 timeout 300 python3 $C3 2>/dev/null > $O/c3.json
  python3 -c "import sys,json; d=json.loads(open('$O/c3.json').readline()); print('C3 ctable $1 diff $2 marks $3:', round(d['kernel_ms'],3), [round(x,2) for x in d['all_kernel_ms']])"
done
done
