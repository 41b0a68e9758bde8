#!/bin/bash
# round 5: the whole GPU suite as the driver runs it (wall clock, slowest tests, binaries' sha256), then the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_suite
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r5_suite/suite.log 2>&1
tail -n 45 gpurun_out/r5_suite/suite.log | cut -c1-200
( time python bench.py ) > gpurun_out/r5_suite/bench.json 2> gpurun_out/r5_suite/bench.err
tail -n 4 gpurun_out/r5_suite/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_suite/bench.json").read().strip().split("\n")[0])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus")})
print("roofline", {k: v for k, v in d["roofline"].items() if k in ("frac", "kernel_ms", "c3_frac_kernel", "c3_kernel_ms", "c5_frac_kernel", "c5_kernel_ms", "traffic")})
print("cpu_baseline", d.get("cpu_baseline"))
sl = d.get("sample_level", {})
print("sample_level", {k: (round(v / 1e6, 1) if isinstance(v, float) and v > 1e5 else v) for k, v in sl.items() if "reads_per_s" in k or "identical" in k or k == "compressed_bytes_per_read"})
c4 = d.get("c4", {})
print("other_k", {k: (round(v["reads_per_s"] / 1e9, 2), v["verify"]["oracle_match"] if v.get("verify") else None) for k, v in ((d.get("c3") or {}).get("other_k") or {}).items()})
print("c4", {k: c4.get(k) for k in ("genotype_wall_s", "host_thread_seconds_per_sample", "counting_wall_s_per_sample", "hmm_device_recursion_s_per_sample", "procs", "host_memory")})
print("bloom", (d.get("bloom") or {}).get("value"), "verify", d.get("verify"), "c3 verify", (d.get("c3") or {}).get("verify"), "c5 verify", (d.get("c5") or {}).get("verify"))
PY
