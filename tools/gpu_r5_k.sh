#!/bin/bash
# round 5: the scan form of the even-k debit pass (parity, rates) and counting streams at the device's highest priority (the c4 run, A/B)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5k
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "even or other_odd or saturation" > gpurun_out/r5k/pytest.log 2>&1
tail -n 6 gpurun_out/r5k/pytest.log | cut -c1-200
python tools/bench_k.py --ks 27,20,22,24,21 > gpurun_out/r5k/bench_k.jsonl 2> gpurun_out/r5k/bench_k.err
cat gpurun_out/r5k/bench_k.jsonl | cut -c1-400; tail -3 gpurun_out/r5k/bench_k.err
for v in 1 0 1 0; do
  VGMI_COUNT_PRIORITY=$v python bench.py --no-c3 --no-c5 --no-bloom --no-sample-level --no-cpu-baseline --steps 3 --reads 20000000 > gpurun_out/r5k/c4_prio$v.json 2> gpurun_out/r5k/c4_prio$v.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5k/c4_prio%s.json" % sys.argv[1]).read().strip().split("\n")[-1])
c4 = d["c4"]
print("priority", sys.argv[1], {k: c4.get(k) for k in ("genotype_wall_s", "counting_wall_s_per_sample", "genotyping_wall_s_per_sample", "hmm_device_recursion_s_per_sample")}, "procs", c4.get("procs", {}).get("genotype_wall_s"))
PY
done
