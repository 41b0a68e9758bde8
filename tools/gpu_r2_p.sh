cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2p
rm -rf $OUT; mkdir -p $OUT
export VGMI_XPART=1
ARGS="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 2"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r2 -- python3 $ARGS > $OUT/b0.json 2> $OUT/e0.log
python3 tools/rocprof_summary.py $OUT/trace | grep -E "calls|xscan|xprobe|count27x"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d $OUT/p1 -o r2 -- python3 $ARGS > $OUT/b1.json 2> $OUT/e1.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d $OUT/p2 -o r2 -- python3 $ARGS > $OUT/b2.json 2> $OUT/e2.log
python3 - <<'PY'
import sqlite3,glob
for db in sorted(glob.glob('gpurun_out/r2p/p*/**/*_results.db',recursive=True)):
    cur=sqlite3.connect(db).cursor()
    for kn,cn,nd,s in cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name"):
        if 'xscan' in kn or 'xprobe' in kn: print(f"{kn[5:18]:14s} {cn:28s} per_dispatch {s/max(nd,1):16.1f}  ({nd})")
PY
find $OUT -name "*.db" -delete
