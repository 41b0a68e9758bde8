# round-2 evidence, second part: ingest against chunk size, device HMM recursion, chr20-scale end to end (one and eight samples)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2final2; rm -rf $OUT; mkdir -p $OUT
timeout 900 python tools/bench_chunk.py 8000000 64,100,256 > $OUT/bench_chunk.jsonl 2> $OUT/bench_chunk.err
VGH_TIMING=1 timeout 2000 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads 10 > $OUT/e2e_chr20_native.json 2> $OUT/e2e.err
VGH_TIMING=1 timeout 2000 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads 10 --gz > $OUT/e2e_chr20_native_gz.json 2>> $OUT/e2e.err
timeout 900 python tools/bench_hmm.py 1000 > $OUT/bench_hmm.json 2>> $OUT/e2e.err
for cfgs in "10 0" "16 0" "16 0,0,0,0"; do set -- $cfgs; timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads $1 --samples 8 --gpus $2 >> $OUT/e2e_chr20_8samples.jsonl 2>> $OUT/e2e.err; done
VGH_HMM_DEVICE=0 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads 16 --samples 8 --gpus 0,0,0,0 >> $OUT/e2e_chr20_8samples.jsonl 2>> $OUT/e2e.err
cat $OUT/bench_chunk.jsonl; cat $OUT/bench_hmm.json; python3 -c "
import json
for l in open('$OUT/e2e_chr20_8samples.jsonl'): d=json.loads(l); print('8 samples', d['threads'], d.get('native_cli_genotype_s'))
"; cut -c1-300 $OUT/e2e_chr20_native.json; python3 -c "
import json
for f in ('e2e_chr20_native.json','e2e_chr20_native_gz.json'):
    d=json.load(open('$OUT/'+f)); print(f, d.get('native_cli_genotype_s'), d.get('native_construct_s')); print('\n'.join(d.get('native_cli_log_tail',[])[-4:]))
"
