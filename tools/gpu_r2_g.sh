cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2g
rm -rf $OUT; mkdir -p $OUT
( time timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.txt
echo "bench rc=$?"; cat $OUT/time.txt; tail -3 $OUT/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2g/bench.json').read())
print(json.dumps(d.get('sample_level'),indent=1))
print('value',d['value'],'frac',d['roofline']['frac'],'c3',d['c3']['value'],d['c3']['roofline']['frac'])
PY
