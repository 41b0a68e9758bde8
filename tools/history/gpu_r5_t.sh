#!/bin/bash
# round 5: the HMM recursion with the products made by wavefronts of their own (VGMI_HMM_SPLIT=1; the kernel lives in the history only:
# commit "HMM recursion with the products made by wavefronts of their own", taken out after this measurement -- DESIGN.md section 9):
# parity, then the CLI's eight-sample and one-sample runs against the shipped kernel, same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5t
VGMI_HMM_SPLIT=1 python -m pytest tests/test_gpu_hmm.py tests/test_gpu_integration.py -q -m gpu -x > gpurun_out/r5t/pytest.log 2>&1
tail -n 3 gpurun_out/r5t/pytest.log | cut -c1-200
for v in 0 1 0 1; do
  for ns in 8 1; do
    VGMI_HMM_SPLIT=$v VG_BENCH_C4_LOG=1 python bench.py --no-c3 --no-c5 --no-bloom --no-sample-level --no-cpu-baseline --steps 3 --reads 20000000 --c4-samples $ns > gpurun_out/r5t/c4_split${v}_$ns.json 2> gpurun_out/r5t/c4_split${v}_$ns.err
    python - "$v" "$ns" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5t/c4_split%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().split("\n")[-1])
c4 = d["c4"]
print("split", sys.argv[1], "samples", sys.argv[2], {k: c4.get(k) for k in ("genotype_wall_s", "counting_wall_s_per_sample", "genotyping_wall_s_per_sample", "hmm_device_recursion_s_per_sample")})
PY
  done
done
