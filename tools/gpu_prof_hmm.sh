# kernel times of the device HMM (recursion, posterior) under rocprofv3, at the shape tools/bench_hmm.py builds
# (synthetic scores: nothing is negligible there, the kernel's worst case; real scores: tools/gpu_e2e_hmm.sh)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hmmprof; rm -rf $OUT; mkdir -p $OUT
python3 tools/bench_hmm.py ${1:-1000} > $OUT/bench_hmm.json
rocprofv3 --kernel-trace --stats -d $OUT/trace -o hmm -- python3 tools/bench_hmm.py ${1:-1000} > $OUT/prof.log 2>&1
cat $OUT/bench_hmm.json
python3 - <<'PY' | tee $OUT/kernel_stats.txt
import glob, sqlite3
db = glob.glob("gpurun_out/hmmprof/trace/*.db")[0]
print("kernel, calls, total us, average us, percent")
for r in sqlite3.connect(db).execute("select * from top_kernels"):
    print(r)
PY
find $OUT -name "*.db" -delete
