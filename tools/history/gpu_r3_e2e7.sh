# one set of chr20-scale files (6 M pairs per sample): eight samples and one sample through `varigraph-mi genotype` as shipped, and
# against the round's knobs (tools/bench_e2e.py --matrix; every run's VCFs are compared with the first run's)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e7; rm -rf $OUT; mkdir -p $OUT
M="${1:-samples=8,t=10;samples=8,t=16;samples=8,t=10,VGH_HMM_CONSUMERS=2;samples=8,t=10,VGH_HMM_CONSUMERS=2,VGH_CPU_BUDGET=0,VGH_DEVICE_GRAPH2NODE=0;samples=1,t=10;samples=1,t=10,VGH_DEVICE_GRAPH2NODE=0}"
VGH_TIMING=1 timeout 2400 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads 10 --gpus 0 --repeat 2 --matrix "$M" > $OUT/matrix.json 2> $OUT/err.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3e2e7/matrix.json"))
for r in d.get("matrix", []):
    print(f"{r['config']:90s} {r['genotype_s']:.2f} s  done in {r.get('done_in')}  loaded {r.get('loaded')}  same VCF {r.get('same_as_first_run')} {r.get('error','')}")
for r in d.get("matrix", [])[:1] + [x for x in d.get("matrix", []) if x["config"] == "samples=1,t=10"][:1]:
    for x in r.get('log', []):
        if "HMM part" not in x: print('    ', x[:250])
PY
