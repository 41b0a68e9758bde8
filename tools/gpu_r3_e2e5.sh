# eight chr20-scale samples in one run with the consumers on one budget of running threads (vgh::CpuBudget), by consumers and
# HMM workgroup packing; then one sample (the first sample's reads counted while the graph loads)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e5; rm -rf $OUT; mkdir -p $OUT
for cfgs in "10 2 0 1" "10 3 0 1" "10 4 0 1" "10 4 1 1" "10 3 1 1" "10 2 0 0" "16 4 1 1" "16 3 0 1"; do set -- $cfgs; VGH_CPU_BUDGET=$4 VGMI_HMM_PACK=$3 VGH_HMM_CONSUMERS=$2 VGH_TIMING=1 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads $1 --samples 8 --gpus 0 > $OUT/o.json 2>> $OUT/e2e.err; python3 -c "
import json
d=json.load(open('$OUT/o.json')); d['consumers']=$2; d['pack']=$3; d['budget']=$4; open('$OUT/e2e_chr20_8samples.jsonl','a').write(json.dumps(d)+'\n'); print('8 samples -t $1 consumers $2 pack $3 budget $4:', d.get('native_cli_genotype_s')); print('\n'.join(x for x in d['native_cli_log_tail'] if 'done in' in x)[-400:])
"; done
for early in 1 0; do VGH_EARLY_COUNT=$early VGH_TIMING=1 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 12000000 --threads 10 --gpus 0 > $OUT/one_$early.json 2>> $OUT/e2e.err; python3 -c "
import json
d=json.load(open('$OUT/one_$early.json')); print('one sample early=$early:', d.get('native_cli_genotype_s')); print('\n'.join(x for x in d['native_cli_log_tail'] if 'done in' in x or 'counting' in x or 'graph loaded' in x)[-600:])
"; done
