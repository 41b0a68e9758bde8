cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2e
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --durations=4 > $OUT/pytest_ingest.log 2>&1
echo "pytest rc=$?"; tail -40 $OUT/pytest_ingest.log
timeout 1500 python tools/bench_pipeline.py 8000000 > $OUT/pipeline.json 2> $OUT/pipeline.err
echo "rc=$?"; python3 -c "
import json
d=json.load(open('$OUT/pipeline.json'))
for k,v in d.items():
    if 'reads_per_s' in k: print(f'{k:45s} {v:.3e}')
"; tail -3 $OUT/pipeline.err
