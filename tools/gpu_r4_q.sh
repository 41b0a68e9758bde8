#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4w6; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
VGMI_INFLATE_WIDE=0 timeout 900 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu > $OUT/tests0.log 2>&1; echo "tests (old batches) rc=$?"; tail -3 $OUT/tests0.log
timeout 600 python3 tools/bench_gzip_only.py 8000000 4 4 > $OUT/gz_t4.json 2> $OUT/gz_t4.err; head -1 $OUT/gz_t4.json
timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 512 > $OUT/bgzf.json 2> $OUT/bgzf.err; tail -1 $OUT/bgzf.json
rocprofv3 --kernel-trace --stats -d $OUT/ktg -o r -- python3 tools/bench_gzip_only.py 4000000 4 4 > $OUT/ktg.json 2> $OUT/ktg.log
rocprofv3 --kernel-trace --stats -d $OUT/ktb -o r -- python3 tools/bench_bgzf_only.py 4000000 4 512 > $OUT/ktb.json 2> $OUT/ktb.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep -i "inflate\|gz_" $OUT/summary.txt | cut -c1-120 | head -8
