# eight chr20-scale samples in one run: HMM workgroups packed two per CU (VGMI_HMM_PACK=1) against one per CU, by consumers
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e4; rm -rf $OUT; mkdir -p $OUT
for cfgs in "10 2 0" "10 4 1" "10 4 0" "16 4 1" "10 8 1" "16 8 1" "10 2 1"; do set -- $cfgs; VGMI_HMM_PACK=$3 VGH_HMM_CONSUMERS=$2 VGH_TIMING=1 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads $1 --samples 8 --gpus 0 > $OUT/o.json 2>> $OUT/e2e.err; python3 -c "
import json
d=json.load(open('$OUT/o.json')); d['consumers']=$2; d['pack']=$3; open('$OUT/e2e_chr20_8samples.jsonl','a').write(json.dumps(d)+'\n'); print('8 samples -t $1 consumers $2 pack $3:', d.get('native_cli_genotype_s')); print('\n'.join(x for x in d['native_cli_log_tail'] if 'done in' in x or 'recursion on the device:' in x)[-400:])
"; done
