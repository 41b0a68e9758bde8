# chr20-scale CLI runs after the emission scores moved to the device: one sample with the phase table, eight samples at -t 10 / -t 16
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e2; rm -rf $OUT; mkdir -p $OUT
VGH_TIMING=1 timeout 2000 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 12000000 --threads 10 > $OUT/e2e_chr20_native.json 2> $OUT/e2e.err
for cfgs in "10 0" "16 0" "16 0,0"; do set -- $cfgs; VGH_TIMING=1 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads $1 --samples 8 --gpus $2 >> $OUT/e2e_chr20_8samples.jsonl 2>> $OUT/e2e.err; done
VGH_HMM_EMIT_DEVICE=0 VGH_TIMING=1 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads 10 --samples 8 --gpus 0 >> $OUT/e2e_chr20_8samples.jsonl 2>> $OUT/e2e.err
python3 -c "
import json
d=json.load(open('$OUT/e2e_chr20_native.json')); print(d.get('native_cli_genotype_s')); print('\n'.join(d.get('native_cli_log_tail',[])[-12:]))
for l in open('$OUT/e2e_chr20_8samples.jsonl'): d=json.loads(l); print('8 samples -t', d['threads'], d.get('native_cli_genotype_s')); print('\n'.join(x for x in d['native_cli_log_tail'] if 'done in' in x or 'emissions on the device' in x)[-400:])
"
