# round-2 GPU call A: memory microbenchmark, the whole GPU suite, bench line, kernel trace (C2 + C3)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT
timeout 600 tools/bin/ubench_mem 32768 > $OUT/ubench_mem.json 2> $OUT/ubench_mem.err
echo "ubench rc=$?"
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cat $OUT/bench.json | cut -c1-1500
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r2 -- python3 bench.py --steps 20 --c3-steps 6 --no-cpu-baseline --verify-reads 0 > $OUT/bench_traced.json 2> $OUT/trace.err
echo "trace rc=$?"
python3 tools/rocprof_summary.py $OUT/trace > $OUT/trace_summary.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
head -30 $OUT/trace_summary.txt
