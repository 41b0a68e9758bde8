#!/bin/bash
# round 5: BASELINE config 5 through the CLI with a real feed: block-gzip files of one tetraploid sample's read pairs (device
# generator -> FASTQ records -> 64 KiB gzip members compressed by a pool of host threads), three sample names over them, back to back
# through ONE `varigraph-mi genotype --sample-ploidy 4 --use-depth` on one GPU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_wgs
PAIRS=${PAIRS:-300000000}
df -h /tmp | tail -1
AVAIL_GB=$(df --output=avail -BG /tmp | tail -1 | tr -dc '0-9')
NEED_GB=$(( PAIRS / 1000000 * 150 / 1000 + 40 ))      # ~0.14 GB of block-gzip text per million pairs (both mates), graph + cohort files
if [ "$AVAIL_GB" -lt "$NEED_GB" ]; then PAIRS=$(( (AVAIL_GB - 40) * 1000 / 150 * 1000000 )); echo "scaled to $PAIRS pairs ($AVAIL_GB GB free)"; fi
python tools/wgs_cli_e2e.py --genome 3000000000 --contigs 24 --variants 5000000 --pairs $PAIRS --files --bgzf --samples 3 --threads 16 \
  > gpurun_out/r5_wgs/e2e.json 2> gpurun_out/r5_wgs/e2e.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_wgs/e2e.json").read().strip().split("\n")[-1])
print({k: d.get(k) for k in ("error", "pairs", "samples", "fastq_files_s", "fastq_file_bytes", "construct_s", "genotype_wall_s", "genotype_peak_rss_gb",
                             "counting_wall_s_per_sample", "counting_reads_per_s", "genotyping_wall_s_per_sample", "dosage_concordance", "prefix_counters_equal_oracle")})
for ln in d.get("genotype_log", [])[-40:]:
    print(ln[:300])
PY
tail -5 gpurun_out/r5_wgs/e2e.err
