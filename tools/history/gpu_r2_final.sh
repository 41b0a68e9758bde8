# round-2 evidence on the final build: bench line, kernel trace of the same command, block-gzip / plain ingest, k != 27
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2final
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r2 -- python3 bench.py --steps 20 --c3-steps 6 --no-cpu-baseline --no-sample-level --verify-reads 0 > $OUT/bench_traced.json 2> $OUT/trace.err
python3 tools/rocprof_summary.py $OUT/trace > $OUT/trace_summary.txt 2>&1
timeout 900 python tools/bench_k.py > $OUT/bench_k.jsonl 2> $OUT/bench_k.err
timeout 900 python tools/bench_pipeline.py 8000000 > $OUT/pipeline.json 2> $OUT/pipeline.err
for cfg in "1200000000 2000000 40000000" "3000000000 5000000 100000000"; do
  set -- $cfg
  timeout 1500 python tools/bench_large.py --genome $1 --variants $2 --reads $3 --steps 3 --check 500000 >> $OUT/bench_large.jsonl 2>> $OUT/bench_large.err
done
find $OUT -name "*.db" -delete
head -12 $OUT/trace_summary.txt; cat $OUT/bench_k.jsonl; cut -c1-400 $OUT/bench_large.jsonl
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read())
print('value',d['value'],'frac',d['roofline']['frac'],'ms',d['ms_per_step'])
print('c3',d['c3']['value'],d['c3']['roofline'])
print(d['sample_level'])
print(d['cpu_baseline']['value'], d['verify'])
"
