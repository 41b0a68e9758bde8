# chr20-scale genotype runs with the device HMM's phase times (VGMI_HMM_TIMING): the recursion kernel on real scores.
# One data set (tools/bench_e2e.py --keep), then the CLI again per VGMI_DBG value in $DBGS (kernel ablations) -- the VCF of
# the DBG=0 run is compared with the first run's; likewise per wavefront count in $WAVES (VGMI_HMM_WAVES).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=/tmp/vg_e2e_keep; rm -rf $W; mkdir -p gpurun_out
VGMI_HMM_TIMING=1 VGH_TIMING=1 timeout 900 python tools/bench_e2e.py --native-only --keep $W --genome 60000000 --variants 500000 --pairs 6000000 --threads 10 > gpurun_out/e2e_t.json 2> gpurun_out/e2e_t.err
python3 -c "
import json
d=json.load(open('gpurun_out/e2e_t.json')); print('genotype_s', d.get('native_cli_genotype_s')); print('\n'.join(l for l in d.get('native_cli_log_tail',[]) if 'vgmi]' in l or 'HMM' in l))
"
cd $W/native_cli && cp sample0.varigraph.vcf.gz first.vcf.gz
for w in ${WAVES:-}; do
  echo "VGMI_HMM_WAVES=$w"
  VGMI_HMM_WAVES=$w VGMI_HMM_TIMING=1 "${GRAFT_REPO_ROOT:-/root/repo}"/varigraph_amd/bin/varigraph-mi genotype --load-graph $W/graph_native.bin -s samples.cfg -t 10 --gpus 0 2>&1 | grep "vgmi\]" | cut -c1-140; cmp sample0.varigraph.vcf.gz first.vcf.gz && echo same VCF
done
for d in ${DBGS:-}; do
  echo "VGMI_DBG=$d"
  VGMI_DBG=$d VGMI_HMM_TIMING=1 "${GRAFT_REPO_ROOT:-/root/repo}"/varigraph_amd/bin/varigraph-mi genotype --load-graph $W/graph_native.bin -s samples.cfg -t 10 --gpus 0 2>&1 | grep "vgmi\]"; cmp sample0.varigraph.vcf.gz first.vcf.gz && echo same VCF
done
rm -rf $W
