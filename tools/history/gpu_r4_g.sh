#!/bin/bash
# (ran against commit fec70cf's parent build: VGMI_CT_OPT=1 -- a lane per aligned pair of counters, 64-bit atomics -- was an A/B knob of
# that build only; measured equal, the code is gone.  Kept as the record of how the same-box comparisons were run.)
# same-box A/B: marks per X vs one mark, 32-bit vs paired 64-bit atomics, workgroups per CU -- each configuration twice, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4g
mkdir -p $O
C3="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 4"
C5="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 2"
for rep in 1 2; do
for cfg in "1 0 6" "0 0 6" "1 1 6" "1 0 8" "1 0 4"; do
  set -- $cfg
  VGMI_CT_MARKS=$1 VGMI_CT_OPT=$2 VGMI_CT_WGS=$3 timeout 300 python3 $C3 2>/dev/null > $O/c3.json
  python3 -c "import sys,json; d=json.loads(open('$O/c3.json').readline()); print('C3 marks $1 opt $2 wgs $3:', round(d['kernel_ms'],3), [round(x,2) for x in d['all_kernel_ms']])"
done
done
for cfg in "1 0 6" "0 0 6" "1 1 6" "1 0 6"; do
  set -- $cfg
  VGMI_CT_MARKS=$1 VGMI_CT_OPT=$2 VGMI_CT_WGS=$3 timeout 600 python3 $C5 2>/dev/null > $O/c5.json
  python3 -c "import sys,json; d=json.loads(open('$O/c5.json').readline()); print('C5 marks $1 opt $2 wgs $3:', round(d['kernel_ms'],3), [round(x,2) for x in d['all_kernel_ms']])"
done
