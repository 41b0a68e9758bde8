cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT
timeout 600 tools/bin/ubench_mem2 > $OUT/ubench_mem2.json 2> $OUT/ubench_mem2.err
echo "ubench2 rc=$?"
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -30 $OUT/pytest_gpu.log
