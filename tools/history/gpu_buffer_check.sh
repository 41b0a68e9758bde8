cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=tests/golden/cohort_sv; W=/tmp/vg_buf; rm -rf $W; mkdir -p $W; gunzip -c $D/graph.bin.gz > $W/graph.bin
export VGH_RANDOM_DEVICE_VALUE=20241022
for b in 1 2 64 3000; do mkdir -p $W/b$b; ( cd $W/b$b; echo "s $OLDPWD/$D/reads_1.fq.gz $OLDPWD/$D/reads_2.fq.gz" > samples.cfg; timeout 300 "$OLDPWD"/varigraph_amd/bin/varigraph-mi genotype --load-graph $W/graph.bin -s samples.cfg -t 4 --buffer $b > log 2>&1; echo "buffer $b rc $?"; ); done
for b in 2 64 3000; do cmp <(gunzip -c $W/b1/s.varigraph.vcf.gz) <(gunzip -c $W/b$b/s.varigraph.vcf.gz) > /dev/null && echo "buffer $b same as 1" || echo "buffer $b DIFFERENT"; done
"$PWD"/varigraph_amd/bin/varigraph-mi genotype --load-graph $W/graph.bin -s $W/b1/samples.cfg --buffer 0 > /dev/null 2>&1; echo "buffer 0 rc $?"
rm -rf $W
