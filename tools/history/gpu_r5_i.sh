#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5i
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_hmm.py -x -q -m gpu --durations=3 > gpurun_out/r5i/hmm.log 2>&1
echo "hmm rc=$?" >> gpurun_out/r5i/hmm.log; tail -6 gpurun_out/r5i/hmm.log | cut -c1-200
VG_TEST_NO_EARLY=1 python -m pytest tests/test_gpu_configs.py tests/test_gpu_integration.py -x -q -m gpu -k "tetra or c5 or ploid or option_combinations or three_different" --durations=6 > gpurun_out/r5i/poly.log 2>&1
echo "poly rc=$?" >> gpurun_out/r5i/poly.log; tail -12 gpurun_out/r5i/poly.log | cut -c1-200
python tools/wgs_cli_e2e.py --genome 3000000000 --contigs 24 --variants 5000000 --pairs 20000000 > gpurun_out/r5i/wgs_20m.json 2> gpurun_out/r5i/wgs.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5i/wgs_20m.json").read().strip().split("\n")[-1])
print({k: d.get(k) for k in ("error", "construct_s", "genotype_wall_s", "genotype_peak_rss_gb", "dosage_concordance", "carrier_concordance", "prefix_counters_equal_oracle", "sites_called", "sites")})
for ln in d.get("genotype_log", [])[-24:]:
    print(ln[:300])
PY
