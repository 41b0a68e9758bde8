# device HMM recursion by the number of chains in ONE launch and the workgroup layout (VGMI_HMM_TIMING prints the kernel's time):
# default = 4 wavefronts per chain, a CU each, up to 64 chains, 2 wavefronts and as many chains per CU as fit beyond;
# VGMI_HMM_PACK=1 = 2 wavefronts per chain, exactly two chains per CU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in 30 60 120 240 480; do
for cfg in "VGMI_HMM_PACK=0" "VGMI_HMM_PACK=1" "VGMI_HMM_WAVES=4"; do
  echo "windows $w $cfg: $(env VGMI_HMM_TIMING=1 $cfg python3 tools/bench_hmm.py 1000 $w 2>&1 | grep -o 'recursion [0-9.]* ms' | sort -k2 -n | head -1)"
done; done
