#!/bin/bash
# round 6: the CLI's count passes with the counter updates in the row loop and deferred, from the smallest deferred block size up: eight
# chr20 samples (12 M pairs) on one GPU, the command's wall and the per-sample "counting ... (kernel ...)" lines of its log
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6_d; rm -rf $OUT; mkdir -p $OUT
D=/tmp/vg_c4_files; rm -rf $D
RUN=$(python3 tools/make_c4_dataset.py $D --pairs 12000000 2> $OUT/make.err | tail -1)
cd $RUN
for cfg in "VGMI_CT_DEFER=0" "VGMI_CT_DEFER=1" "VGMI_CT_DEFER=1 VGMI_CT_DEFER_MIN=67108864" "VGMI_CT_DEFER=1 VGMI_CT_DEFER_MIN=268435456" "VGMI_CT_DEFER=0" "VGMI_CT_DEFER=1"; do
  for rep in 1 2; do
    /usr/bin/env $cfg VGH_RANDOM_DEVICE_VALUE=20241022 VGH_TIMING=1 "$OLDPWD/varigraph_amd/bin/varigraph-mi" genotype --load-graph $D/graph.bin -s samples.cfg -t 10 --gpus 0 > /dev/null 2> $OLDPWD/$OUT/cli.err
    echo "$cfg: $(grep 'done in' $OLDPWD/$OUT/cli.err | sed 's/.*done in//') | kernel s per sample: $(grep -o '(kernel [0-9.]* s' $OLDPWD/$OUT/cli.err | awk '{s+=$2; n++} END {printf "%.4f (n=%d)", s/n, n}') | counting: $(grep -o 'counting [0-9.]* s' $OLDPWD/$OUT/cli.err | awk '{s+=$2; n++} END {printf "%.3f", s/n}')"
  done
done | tee $OLDPWD/$OUT/cli_defer.txt
rm -rf $D
