# round-3 evidence: chr20-scale end to end through the CLI (one sample with the per-phase table, eight samples), the tetraploid
# 30 Mb cohort and the reference alongside (byte-identical outputs), on the round's final build
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3e2e; rm -rf $OUT; mkdir -p $OUT
VGH_TIMING=1 timeout 2000 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 12000000 --threads 10 > $OUT/e2e_chr20_native.json 2> $OUT/e2e.err
for cfgs in "10 0" "16 0"; do set -- $cfgs; VGH_TIMING=1 timeout 1500 python tools/bench_e2e.py --native-only --genome 60000000 --variants 500000 --pairs 6000000 --threads $1 --samples 8 --gpus $2 >> $OUT/e2e_chr20_8samples.jsonl 2>> $OUT/e2e.err; done
timeout 2400 python tools/bench_e2e.py --genome 60000000 --variants 500000 --pairs 12000000 --threads 10 > $OUT/e2e_chr20_full.json 2>> $OUT/e2e.err
timeout 900 python tools/bench_e2e.py --genome 30000000 --variants 100000 --pairs 2000000 --threads 10 --ploidy 4 --vcf-samples 3 --indel 0.05 --sv 0.001 > $OUT/e2e_30mb_tetraploid_full.json 2>> $OUT/e2e.err
python3 -c "
import json
for f in ('e2e_chr20_native.json','e2e_chr20_full.json','e2e_30mb_tetraploid_full.json'):
    d=json.load(open('$OUT/'+f)); print(f, {k:d.get(k) for k in ('native_cli_genotype_s','native_construct_s','reference_cpu_genotype_s','reference_construct_s','vcf_identical','graph_identical')}); print('\n'.join(d.get('native_cli_log_tail',[])[-8:]))
for l in open('$OUT/e2e_chr20_8samples.jsonl'): d=json.loads(l); print('8 samples -t', d['threads'], d.get('native_cli_genotype_s'))
"
