# eight chr20-scale samples in one run against the number of HMM consumers (emission scores on the device)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e3; rm -rf $OUT; mkdir -p $OUT
for cfgs in "10 2" "10 4" "16 2" "16 4"; do set -- $cfgs; VGH_HMM_CONSUMERS=$2 VGH_TIMING=1 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads $1 --samples 8 --gpus 0 > $OUT/o.json 2>> $OUT/e2e.err; python3 -c "
import json
d=json.load(open('$OUT/o.json')); d['consumers']=$2; open('$OUT/e2e_chr20_8samples.jsonl','a').write(json.dumps(d)+'\n'); print('8 samples -t $1 consumers $2:', d.get('native_cli_genotype_s')); print('\n'.join(x for x in d['native_cli_log_tail'] if 'done in' in x or 'emissions on the device:' in x)[-500:])
"; done
