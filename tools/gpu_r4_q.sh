#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gunzip.py tests/test_gpu_ingest.py -q > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -4 $O/t.log | cut -c1-200
timeout 600 python3 tools/bench_gzip_only.py 4000000 4 2,16 2>$O/e.log | cut -c1-140
timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 256 2>>$O/e.log | cut -c1-140
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 tools/bench_gunzip.py 2000000 4 > $O/b.json 2> $O/b.err
python3 tools/rocprof_summary.py $O > $O/summary.txt; find $O -name "*.db" -delete
grep "gz_\|calls" $O/summary.txt | cut -c1-120
