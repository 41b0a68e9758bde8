#!/bin/bash
# round 5: the FASTQ parser's small kernels (tiles of 16 KiB, scans that fetch together, no work behind the last record): ingest tests,
# then the sample-level rates and the eight-sample CLI run of the bench, twice
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5r
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_gunzip.py tests/test_gpu_integration.py -q -m gpu > gpurun_out/r5r/pytest.log 2>&1
tail -n 3 gpurun_out/r5r/pytest.log | cut -c1-200
for i in 1 2; do
  VG_BENCH_C4_LOG=1 python bench.py --no-c3 --no-c5 --no-bloom --no-cpu-baseline --steps 3 --reads 20000000 > gpurun_out/r5r/bench_$i.json 2> gpurun_out/r5r/bench_$i.err
  python - "$i" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5r/bench_%s.json" % sys.argv[1]).read().strip().split("\n")[-1])
c4 = d["c4"]
print({k: c4.get(k) for k in ("genotype_wall_s", "counting_wall_s_per_sample", "genotyping_wall_s_per_sample", "hmm_device_recursion_s_per_sample")}, "procs", c4.get("procs", {}).get("genotype_wall_s"))
sl = d.get("sample_level", {})
print({k: round(v / 1e6, 1) for k, v in sl.items() if "reads_per_s" in k and isinstance(v, float)})
PY
done
