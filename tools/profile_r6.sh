#!/bin/bash
# Round-6 profiling recipe for the count kernels on the CURRENT build: kernel-trace stats, then PMC passes (counters only, one --pmc set
# per run, the program directly behind `--`), then profiles/hbm_traffic*.json regenerated WITH the sha256 of libvgmi.so and the kernel's
# name (bench.py's measured_traffic() only accepts a file whose sha256 and kernel match the library it runs).
# Usage: tools/profile_r6.sh [c3] [c5] [c2] [bloom]      (env such as VGMI_CT_DEFER is inherited: the tag of the output is $TAG, default r6)
#   bloom: K3's binned form (the 60 Mb call of the bench line's bloom block): rows_kernel<KEYS> + bb_scatter1/2 + bb_accumulate
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r6}
for W in ${*:-c3 c5 c2}; do
  OUT=gpurun_out/prof_${TAG}_$W
  rm -rf $OUT; mkdir -p $OUT
  case $W in
    c3) A="tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3"; N=24000000 ;;
    c5) A="tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 1"; N=100000000 ;;
    c2) A="bench.py --no-c3 --no-c5 --no-bloom --no-sample-level --no-c4 --no-cpu-baseline"; N=100000000 ;;
    bloom) A="tools/bench_bloom.py --genome 60000000 --steps 3"; N=59999974 ;;
  esac
  RX='count27|countkc|ctd_'; [ $W = bloom ] && RX='bb_|rows_kernel' 
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r6 -- python3 $A > $OUT/b0.json 2> $OUT/e0.log
  i=0
  for pm in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    { [ $W = c2 ] || [ $W = bloom ]; } && [ $i -gt 2 ] && break      # C2: bytes only (the instruction-mix counters of this kernel: tools/pmc_c2.sh)
    timeout 900 rocprofv3 --kernel-include-regex "$RX" --pmc $pm -d $OUT/pmc_$i -o r6 -- python3 $A > $OUT/b$i.json 2> $OUT/e$i.log
  done
  python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
  find $OUT -name "*.db" -delete
  sha256sum varigraph_amd/libvgmi.so | cut -d' ' -f1 > $OUT/libvgmi.sha256
  python3 -c "from varigraph_amd import build; print(build.source_digest())" > $OUT/source.sha256
  python3 tools/make_traffic_json_r6.py $W $N $OUT "$RX"
  grep -v "^$" $OUT/summary.txt | grep "$RX\|PMC\|kernel-trace\|calls" | cut -c1-160 | head -40
done
