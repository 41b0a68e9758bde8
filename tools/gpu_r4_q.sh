#!/bin/bash
# (scratch) K3 binned: parity, then A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4b; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bloom or mbf" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -6 $OUT/tests.log
for b in 1 0 1 0; do VGMI_BLOOM_BINNED=$b timeout 300 python3 tools/bench_bloom.py --genome 60000000 --steps 3 > $OUT/bloom_b$b.json 2> $OUT/bloom_b$b.err; echo "binned=$b"; tail -1 $OUT/bloom_b$b.json; tail -2 $OUT/bloom_b$b.err; done
