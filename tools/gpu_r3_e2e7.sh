cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e7; rm -rf $OUT; mkdir -p $OUT
M="samples=8,t=10,VGH_HMM_CONSUMERS=4;samples=8,t=10,VGH_HMM_CONSUMERS=8;samples=1,t=10"
VGH_TIMING=1 VGMI_HMM_TIMING=1 timeout 2400 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads 10 --gpus 0 --repeat 1 --matrix "$M" > $OUT/matrix.json 2> $OUT/err.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3e2e7/matrix.json"))
for r in d.get("matrix", []):
    print(f"{r['config']:70s} {r['genotype_s']:.2f} s  done in {r.get('done_in')}  loaded {r.get('loaded')}  same VCF {r.get('same_as_first_run')} {r.get('error','')}")
    for x in r.get('log', []): print('    ', x[:330])
PY
