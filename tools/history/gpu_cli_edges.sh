# the CLI at the corners of its thread / consumer / part arithmetic: one and two threads, three samples, tiny windows;
# every VCF must equal the -t 4 single-sample one (golden cohort_snp)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=tests/golden/cohort_snp; W=/tmp/vg_edges; rm -rf $W; mkdir -p $W; gunzip -c $D/graph.bin.gz > $W/graph.bin
export VGH_RANDOM_DEVICE_VALUE=20241022
run() { # name, then the CLI's options
  n=$1; shift; mkdir -p $W/$n; ( cd $W/$n; for s in a b c; do echo "$s $OLDPWD/$D/reads_1.fq.gz $OLDPWD/$D/reads_2.fq.gz"; done > samples.cfg
  timeout 120 "$OLDPWD"/varigraph_amd/bin/varigraph-mi genotype --load-graph $W/graph.bin -s samples.cfg "$@" > log 2>&1 || { echo "$n FAILED"; tail -3 log; } )
}
run base -t 4
run t1 -t 1
run t2 -t 2 --granularity 0.001
run t3 -t 3 --granularity 0.001 --gpus 0,0,0
VGH_HMM_DEVICE=0 run host -t 2 --granularity 0.001
for n in t1; do for s in a b c; do cmp <(gunzip -c $W/base/a.varigraph.vcf.gz | sed 's/\ta$/\tS/;s/\tb$/\tS/;s/\tc$/\tS/') <(gunzip -c $W/$n/$s.varigraph.vcf.gz | sed 's/\ta$/\tS/;s/\tb$/\tS/;s/\tc$/\tS/') > /dev/null && echo "$n $s same" || echo "$n $s DIFFERENT"; done; done
for n in t3 host; do for s in a b c; do cmp <(gunzip -c $W/t2/a.varigraph.vcf.gz | sed 's/\ta$/\tS/;s/\tb$/\tS/;s/\tc$/\tS/') <(gunzip -c $W/$n/$s.varigraph.vcf.gz | sed 's/\ta$/\tS/;s/\tb$/\tS/;s/\tc$/\tS/') > /dev/null && echo "$n $s same" || echo "$n $s DIFFERENT"; done; done
rm -rf $W
