cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2j
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r2 -- python3 tools/bench_bgzf_only.py 4000000 4 128 > $OUT/bgzf.json 2> $OUT/bgzf.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $OUT/pmc_sq -o r2 -- python3 tools/bench_bgzf_only.py 4000000 4 128 > $OUT/b1.json 2> $OUT/e1.err
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCP_TCC_WRITE_REQ_sum -d $OUT/pmc_ea -o r2 -- python3 tools/bench_bgzf_only.py 4000000 4 128 > $OUT/b2.json 2> $OUT/e2.err
cat $OUT/bgzf.json
python3 tools/rocprof_summary.py $OUT/trace | grep -E "calls|inflate|count27"
python3 - <<'PY'
import sqlite3,glob
for db in sorted(glob.glob('gpurun_out/r2j/pmc_*/**/*_results.db',recursive=True)):
    cur=sqlite3.connect(db).cursor()
    for kn,cn,nd,s in cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name"):
        if 'inflate' in kn: print(f"{cn:28s} per_dispatch {s/max(nd,1):16.1f}  ({nd} dispatches)")
PY
find $OUT -name "*.db" -delete
