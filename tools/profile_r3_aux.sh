#!/bin/bash
# Round-3 evidence for the kernels off the bench's main line: K3 Bloom update (rows_kernel<2, false>), K4 query
# (bloom_query_kernel), K5 / K6 (cov_kernel, node_gather_kernel): kernel-trace stats + FETCH_SIZE / WRITE_SIZE in their own passes
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_r3_aux
rm -rf $OUT; mkdir -p $OUT
B="python3 tools/bench_bloom.py --genome 60000000 --steps 3 --query 10000000"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r3 -- $B > $OUT/b0.json 2> $OUT/e0.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r3 -- $B > $OUT/b1.json 2> $OUT/e1.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r3 -- $B > $OUT/b2.json 2> $OUT/e2.log
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_ea -o r3 -- $B > $OUT/b3.json 2> $OUT/e3.log
B2="python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-sample-level --no-c3 --no-c5 --no-bloom --verify-reads 0"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_k56_fetch -o r3 -- $B2 > $OUT/b4.json 2> $OUT/e4.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_k56_write -o r3 -- $B2 > $OUT/b5.json 2> $OUT/e5.log
python3 - <<'PY' > $OUT/summary.txt
import glob, os, sqlite3, sys
sys.path.insert(0, "tools")
import rocprof_summary as r
for db in sorted(glob.glob("gpurun_out/prof_r3_aux/**/*_results.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    n = cur.execute("select count(*) from counters_collection").fetchone()[0]
    txt = r.pmc_stats(db, kernel_filter=("rows_kernel<2", "bloom_query", "cov_kernel", "node_gather")) if n else r.kernel_stats(db)
    if txt:
        print(txt)
        print()
PY
find $OUT -name "*.db" -delete
cat $OUT/summary.txt | head -60
