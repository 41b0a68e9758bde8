cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2i
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r2 -- python3 tools/bench_bgzf_only.py 4000000 4 > $OUT/bgzf.json 2> $OUT/bgzf.err
cat $OUT/bgzf.json
python3 tools/rocprof_summary.py $OUT/trace | head -24
find $OUT -name "*.db" -delete
