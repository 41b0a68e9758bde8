# one set of chr20-scale files, the native genotype run per configuration (tools/bench_e2e.py --matrix): eight samples by consumers /
# thread budget / workgroup packing, one sample with and without the device-side graph2node lookups and the early counting
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e6; rm -rf $OUT; mkdir -p $OUT
M="samples=8,t=10,VGH_HMM_CONSUMERS=4"
M="$M;samples=8,t=10,VGH_HMM_CONSUMERS=4,VGMI_HMM_PACK=1"
M="$M;samples=8,t=10,VGH_HMM_CONSUMERS=3"
M="$M;samples=8,t=10,VGH_HMM_CONSUMERS=2"
M="$M;samples=8,t=10,VGH_HMM_CONSUMERS=2,VGH_CPU_BUDGET=0"
M="$M;samples=8,t=10,VGH_HMM_CONSUMERS=4,VGH_DEVICE_GRAPH2NODE=0"
M="$M;samples=8,t=16,VGH_HMM_CONSUMERS=4"
M="$M;samples=8,t=16,VGH_HMM_CONSUMERS=4,VGMI_HMM_PACK=1"
M="$M;samples=8,t=16,VGH_HMM_CONSUMERS=3"
M="$M;samples=1,t=10"
M="$M;samples=1,t=10,VGH_DEVICE_GRAPH2NODE=0"
M="$M;samples=1,t=10,VGH_EARLY_COUNT=0"
M="$M;samples=1,t=10,VGH_EARLY_COUNT=0,VGH_DEVICE_GRAPH2NODE=0"
VGH_TIMING=1 timeout 2400 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads 10 --gpus 0 --repeat 2 --matrix "$M" > $OUT/matrix.json 2> $OUT/err.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3e2e6/matrix.json"))
for r in d.get("matrix", []):
    print(f"{r['config']:70s} {r['genotype_s']:.2f} s  done in {r.get('done_in')}  loaded {r.get('loaded')}  same VCF {r.get('same_as_first_run')} {r.get('error','')}")
PY
