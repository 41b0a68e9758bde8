# kernel times of the device HMM (recursion, posterior) under rocprofv3, at the shape tools/bench_hmm.py builds
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hmmprof; rm -rf $OUT; mkdir -p $OUT
python3 tools/bench_hmm.py ${1:-1000} > $OUT/bench_hmm.json
rocprofv3 --kernel-trace --stats -d $OUT/trace -o hmm -- python3 tools/bench_hmm.py ${1:-1000} > $OUT/prof.log 2>&1
cat $OUT/bench_hmm.json
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats.csv; cut -c1-200 $OUT/kernel_stats.csv | head -8
