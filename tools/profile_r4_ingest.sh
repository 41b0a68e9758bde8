#!/bin/bash
# Round 4, compressed FASTQ on the device: A/B of the two inflate forms on one box (reads/s, files -> counters), then the kernel
# tables of one block-gzip and one gzip run (rocprofv3 --kernel-trace --stats).  Results: gpurun_out/r4_ingest/ (copy
# ingest.jsonl and summary.txt into profiles/ as r4_ingest.jsonl, r4_ingest_rocprofv3_summary.txt).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4_ingest; rm -rf $OUT; mkdir -p $OUT
: > $OUT/ingest.jsonl
for w in 1 0 1 0; do
  VGMI_INFLATE_WIDE=$w timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 512 2> $OUT/bgzf_w$w.err | tail -1 | sed "s/^{/{\"inflate_wide\": $w, /" >> $OUT/ingest.jsonl
done
for w in 1 0; do for t in 2 4; do
  VGMI_INFLATE_WIDE=$w timeout 600 python3 tools/bench_gzip_only.py 8000000 4 $t 2> $OUT/gz_w${w}_t$t.err | sed "s/^{/{\"inflate_wide\": $w, /" >> $OUT/ingest.jsonl
done; done
cat $OUT/ingest.jsonl
rocprofv3 --kernel-trace --stats -d $OUT/bgzf -o r -- python3 tools/bench_bgzf_only.py 4000000 4 512 > $OUT/kt_bgzf.json 2> $OUT/kt_bgzf.log
rocprofv3 --kernel-trace --stats -d $OUT/gzip -o r -- python3 tools/bench_gzip_only.py 4000000 4 4 > $OUT/kt_gzip.json 2> $OUT/kt_gzip.log
# instruction mix and stalls of the two decoders (their own --pmc passes)
rocprofv3 --kernel-include-regex "bgzf_inflate" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_bgzf -o r -- python3 tools/bench_bgzf_only.py 4000000 4 512 > /dev/null 2> $OUT/pmc_bgzf.log
rocprofv3 --kernel-include-regex "gz_decode" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_gzip -o r -- python3 tools/bench_gzip_only.py 4000000 4 4 > /dev/null 2> $OUT/pmc_gzip.log
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
grep -i "inflate\|gz_" $OUT/summary.txt | cut -c1-130 | head -30
