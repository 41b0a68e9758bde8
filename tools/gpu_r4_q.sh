#!/bin/bash
# (scratch) the wide inflate batches with parallel match copies: parity suites, A/B, kernel trace
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4w; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
for w in 1 0 1 0; do
  VGMI_INFLATE_WIDE=$w timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 512 > $OUT/bgzf_w$w.json 2> $OUT/bgzf_w$w.err; echo "bgzf wide=$w"; tail -1 $OUT/bgzf_w$w.json
done
for w in 1 0; do
  VGMI_INFLATE_WIDE=$w timeout 600 python3 tools/bench_gzip_only.py 8000000 4 4 > $OUT/gz_w$w.json 2> $OUT/gz_w$w.err; echo "gzip wide=$w"; head -1 $OUT/gz_w$w.json
done
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 tools/bench_bgzf_only.py 4000000 4 512 > $OUT/kt.json 2> $OUT/kt.log
rocprofv3 --kernel-trace --stats -d $OUT/ktg -o r -- python3 tools/bench_gzip_only.py 4000000 4 4 > $OUT/ktg.json 2> $OUT/ktg.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep -i "inflate\|gz_" $OUT/summary.txt | cut -c1-120
