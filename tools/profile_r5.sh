#!/bin/bash
# Round-5 profiling recipe (one MI355X box).  Usage: tools/profile_r5.sh [c2] [k] [c3] [c5] [hmm] [aux]
#   c2   kernel table of the default bench command's own leg (count27s_kernel<true, 27>) -> r5_rocprofv3_summary.txt
#   k    tools/bench_k.py over k = 19 .. 28 + its kernel table (count27s_kernel<true, K>, even_debit_kernel) -> r5_bench_k.jsonl, r5_k_rocprofv3_summary.txt
#   c3 / c5  kernel table of the chr20-class / whole-genome-class launch (count27c_kernel: unchanged since round 4, whose PMC passes stand)
#   largek  kernel tables of the chr20-class launch at k = 21 and k = 22 (context table at other k; the even-k pass), FETCH_SIZE / WRITE_SIZE of the k = 22 kernels -> r5_largek_rocprofv3_summary.txt
#   hmm  kernel table of `varigraph-mi genotype` on the bench's c4 files (eight chr20-scale samples): the HMM kernels -> r5_hmm_rocprofv3_summary.txt
#   aux  one-counter PMC passes on the kernels round 4 got no rows for (bb_*, gz_decode_kernel), each with --kernel-include-regex -> r5_aux_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W="${*:-c2 k hmm aux}"
for w in $W; do
  OUT=gpurun_out/prof_r5_$w; rm -rf $OUT; mkdir -p $OUT
  case $w in
  c2)
    rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 bench.py --no-c3 --no-c5 --no-bloom --no-sample-level --no-c4 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
    ;;
  k)
    python3 tools/bench_k.py --ks 19,21,23,25,27,20,22,24,26,28 > $OUT/bench_k.jsonl 2> $OUT/bench_k.err
    rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 tools/bench_k.py --ks 21,25,27,22 > $OUT/bench_k_traced.jsonl 2>> $OUT/bench_k.err
    cat $OUT/bench_k.jsonl | cut -c1-170
    ;;
  largek)
    # the chr20-class launch over the graphs of k = 21 (countkc_kernel<21u>) and k = 22 (even: scan + walk + countkc_kernel<22u> + tail), then
    # FETCH_SIZE / WRITE_SIZE of the scan and of the k = 21 kernel in passes of their own
    rocprofv3 --kernel-trace --stats -d $OUT/kt21 -o r -- python3 tools/bench_large.py --k 21 --steps 3 > $OUT/b21.json 2> $OUT/e21.log
    rocprofv3 --kernel-trace --stats -d $OUT/kt22 -o r -- python3 tools/bench_large.py --k 22 --steps 3 > $OUT/b22.json 2> $OUT/e22.log
    for pm in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --kernel-include-regex "even_debit_kernel|countkc_kernel" --pmc $pm -d $OUT/pmc_$pm -o r -- python3 tools/bench_large.py --k 22 --steps 1 > $OUT/pmc_$pm.out 2> $OUT/pmc_$pm.err
      echo "counters '$pm': rc=$?"
    done
    ;;
  c3)
    rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 tools/bench_large.py --genome 60000000 --variants 500000 --reads 24000000 --steps 3 > $OUT/b.json 2> $OUT/e.log
    ;;
  c5)
    rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- python3 tools/bench_large.py --genome 3000000000 --variants 5000000 --reads 100000000 --steps 1 > $OUT/b.json 2> $OUT/e.log
    ;;
  hmm)
    D=/tmp/vg_c4_files; rm -rf $D
    RUN=$(python3 tools/make_c4_dataset.py $D 2> $OUT/make.err | tail -1)
    ( cd $RUN && VGH_ATEXIT=1 VGH_RANDOM_DEVICE_VALUE=20241022 VGH_TIMING=1 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/kt" -o r -- "$OLDPWD/varigraph_amd/bin/varigraph-mi" genotype --load-graph $D/graph.bin -s samples.cfg -t 10 --gpus 0 > "$OLDPWD/$OUT/cli.out" 2> "$OLDPWD/$OUT/cli.err" )
    grep "done in" $OUT/cli.err
    rm -rf $D
    ;;
  aux)
    : > $OUT/aux_pmc.txt
    for spec in "bb_scatter|bb_accumulate|bb_keys@tools/bench_bloom.py" "gz_decode@tools/bench_gzip_only.py 2000000 4 4"; do
      rx="${spec%%@*}"; cmd="${spec#*@}"
      i=0
      for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
        i=$((i+1))
        d=$OUT/pmc_$(echo $rx | tr -dc 'a-z' | cut -c1-12)_$i
        timeout 600 rocprofv3 --kernel-include-regex "$rx" --pmc $pm -d $d -o r -- python3 $cmd > $d.out 2> $d.err
        echo "regex '$rx' counters '$pm' cmd '$cmd': rc=$?" | tee -a $OUT/aux_pmc.txt
      done
    done
    ;;
  esac
  python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
  find $OUT -name "*.db" -delete
  echo "== $w"; head -25 $OUT/summary.txt | cut -c1-170
done
