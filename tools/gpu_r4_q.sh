#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4w12; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for kb in 32 48 40 32; do VGMI_GZ_SEG_KB=$kb timeout 600 python3 tools/bench_gzip_only.py 8000000 4 4 > $OUT/gz_$kb.json 2> $OUT/gz_$kb.err; echo "seg $kb"; head -1 $OUT/gz_$kb.json; done
rocprofv3 --kernel-trace --stats -d $OUT/ktg -o r -- python3 tools/bench_gzip_only.py 4000000 4 4 > $OUT/ktg.json 2> $OUT/ktg.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep -i "gz_" $OUT/summary.txt | cut -c1-120 | head -6
