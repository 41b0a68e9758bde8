# where a chain's time goes: instruction-mix and wait counters of hmm_recursion_kernel at the shape of tools/bench_hmm.py
# (60 chains of 1 000 nodes, 120 genotypes: one workgroup of four wavefronts per CU, a wavefront per SIMD)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_hmm; rm -rf $OUT; mkdir -p $OUT
i=0
for pm in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
          "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $pm -d $OUT/pmc_$i -o hmm -- python3 tools/bench_hmm.py 1000 ${1:-30} > $OUT/bench_$i.json 2> $OUT/err_$i.log
done
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep "hmm_recursion" $OUT/summary.txt
