#!/bin/bash
# round 5: the CLI-level eight-sample run of the bench (its c4 block) with the stage lines of the log kept
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_c4
VG_BENCH_C4_LOG=1 python bench.py --no-c3 --no-c5 --no-bloom --no-sample-level --no-cpu-baseline --steps 5 --reads 20000000 "$@" > gpurun_out/r5_c4/bench.json 2> gpurun_out/r5_c4/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_c4/bench.json").read().strip().split("\n")[-1])
c4 = d["c4"]
log = c4.pop("log", [])
print(json.dumps(c4, indent=1))
print("\n".join(log[-80:]))
PY
