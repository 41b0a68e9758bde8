# kernel trace of the block-gzip sample path (device inflate + parser + count); CHUNKS="kb kb ..." sweeps the text chunk
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for kb in ${CHUNKS:-0}; do
  OUT=gpurun_out/prof_bgzf_$kb; rm -rf $OUT; mkdir -p $OUT
  if [ "$kb" != "0" ]; then export VGMI_FASTQ_CHUNK_KB=$kb; fi
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python3 tools/bench_bgzf_only.py 8000000 4 100 > $OUT/run.json 2> $OUT/err.txt
  python3 tools/rocprof_summary.py $OUT/trace > $OUT/summary.txt 2>&1
  find $OUT -name "*.db" -delete
  echo "chunk_kb=$kb"; grep -m1 bgzf_inflate $OUT/summary.txt | cut -c1-60; cat $OUT/run.json
done
