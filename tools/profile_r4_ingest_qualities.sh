#!/bin/bash
# Round 4: the ingest A/B on compressed FASTQ with binned random qualities (what a real .fastq.gz looks like to the inflate kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4_qual; rm -rf $OUT; mkdir -p $OUT
export VG_BENCH_QUAL=binned
: > $OUT/ingest_binned_qualities.jsonl
for w in 1 0; do
  VGMI_INFLATE_WIDE=$w timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 512 2> $OUT/bgzf_w$w.err | tail -1 | sed "s/^{/{\"qualities\": \"binned\", \"inflate_wide\": $w, /" >> $OUT/ingest_binned_qualities.jsonl
  VGMI_INFLATE_WIDE=$w timeout 600 python3 tools/bench_gzip_only.py 8000000 4 4 2> $OUT/gz_w$w.err | sed "s/^{/{\"qualities\": \"binned\", \"inflate_wide\": $w, /" >> $OUT/ingest_binned_qualities.jsonl
done
cat $OUT/ingest_binned_qualities.jsonl | cut -c1-260
