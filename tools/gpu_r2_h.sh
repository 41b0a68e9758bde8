cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2h
rm -rf $OUT; mkdir -p $OUT
timeout 900 python tools/bench_k.py > $OUT/bench_k.jsonl 2> $OUT/bench_k.err
echo "bench_k rc=$?"; cat $OUT/bench_k.jsonl; tail -3 $OUT/bench_k.err
for cfg in "5 0" "5 1" "6 0" "6 1" "7 1"; do
  set -- $cfg
  VGMI_LOCALITY=$1 VGMI_SLOT_ORDER=$2 timeout 600 python tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 4 --check 500000 > $OUT/chr20_b$1_so$2.json 2> $OUT/chr20_b$1_so$2.err
  echo "chr20 bucket_log2=$1 slot_order=$2: $(cut -c1-20,150-330 $OUT/chr20_b$1_so$2.json)"
done
