cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2c
rm -rf $OUT; mkdir -p $OUT
timeout 900 tools/bin/ubench_mem3 > $OUT/ubench_mem3.json 2> $OUT/ubench_mem3.err
echo "ubench3 rc=$?"; tail -3 $OUT/ubench_mem3.err
