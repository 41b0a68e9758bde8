cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2y
rm -rf $OUT; mkdir -p $OUT
export VGMI_XTABLE=1
for sh in -1 0; do
  VGMI_XTABLE_SHIFT=$sh timeout 600 python tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3 --check 500000 > $OUT/chr20_sh$sh.json 2> $OUT/chr20_sh$sh.err
  echo "chr20 xtable shift=$sh: $(cut -c150-330 $OUT/chr20_sh$sh.json)"
done
timeout 900 python tools/bench_large.py --genome 1200000000 --variants 2000000 --reads 40000000 --steps 3 --check 500000 > $OUT/g12.json 2> $OUT/g12.err
echo "1.2Gb xtable: $(cut -c150-330 $OUT/g12.json)"
timeout 1500 python tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 3 --check 300000 > $OUT/wgs.json 2> $OUT/wgs.err
echo "WGS xtable: $(cut -c150-330 $OUT/wgs.json)"; tail -2 $OUT/wgs.err
ARGS="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 2"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum -d $OUT/p1 -o r2 -- python3 $ARGS > $OUT/b1.json 2> $OUT/e1.log
python3 - <<'PY'
import sqlite3,glob
for db in sorted(glob.glob('gpurun_out/r2y/p*/**/*_results.db',recursive=True)):
    cur=sqlite3.connect(db).cursor()
    for kn,cn,nd,s in cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name"):
        if 'count27x' in kn: print(f"{cn:36s} per_dispatch {s/max(nd,1):16.1f}  ({nd})")
PY
find $OUT -name "*.db" -delete
