cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2e
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --durations=8 > $OUT/pytest_ingest.log 2>&1
echo "pytest rc=$?"; tail -40 $OUT/pytest_ingest.log
