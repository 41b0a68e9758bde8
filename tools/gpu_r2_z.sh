cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2z
rm -rf $OUT; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 1500 python tools/bench_e2e.py --genome 10000000 --variants 80000 --pairs 1000000 > $OUT/e2e_10mb.json 2> $OUT/e2e.err
python3 -c "
import json
d=json.load(open('$OUT/e2e_10mb.json'))
print({k:v for k,v in d.items() if not k.endswith('log_tail') and not k.endswith('_log')})
print('\n'.join(d.get('native_cli_log_tail',[])))
"
