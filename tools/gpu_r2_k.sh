cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2k
rm -rf $OUT; mkdir -p $OUT
for mul in 8 4 2; do
  VGMI_TABLE_MUL=$mul timeout 1500 python tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 3 --check 300000 > $OUT/wgs_mul$mul.json 2> $OUT/wgs_mul$mul.err
  echo "WGS table_mul=$mul: $(cut -c60-400 $OUT/wgs_mul$mul.json)"
done
for mul in 4 2; do
  VGMI_TABLE_MUL=$mul timeout 900 python tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3 --check 300000 > $OUT/chr20_mul$mul.json 2> $OUT/chr20_mul$mul.err
  echo "chr20 table_mul=$mul: $(cut -c60-400 $OUT/chr20_mul$mul.json)"
done
