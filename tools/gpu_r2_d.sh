cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2d
rm -rf $OUT; mkdir -p $OUT
for so in 0 1; do
  VGMI_SLOT_ORDER=$so timeout 600 python tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 4 --check 1000000 > $OUT/chr20_so$so.json 2> $OUT/chr20_so$so.err
  echo "chr20 slot_order=$so: $(cat $OUT/chr20_so$so.json)"
done
for so in 0 1; do
  VGMI_SLOT_ORDER=$so timeout 900 python tools/bench_large.py --genome 1200000000 --variants 2000000 --reads 40000000 --steps 3 --check 500000 > $OUT/g12_so$so.json 2> $OUT/g12_so$so.err
  echo "1.2Gb slot_order=$so: $(cat $OUT/g12_so$so.json)"
done
