cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2f
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python tools/bench_pipeline.py 16000000 > $OUT/pipeline.json 2> $OUT/pipeline.err
echo "rc=$?"; cat $OUT/pipeline.json; tail -5 $OUT/pipeline.err
