# device HMM recursion by the number of chains in ONE launch and the workgroup layout (VGMI_HMM_TIMING prints the kernel's time):
# default = 4 wavefronts per chain, a CU each, up to 64 chains, 2 wavefronts and as many chains per CU as fit beyond;
# (a layout with exactly two chains per CU was tried as VGMI_HMM_PACK=1: 78.7 ms for 960 chains against 39.5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in 30 60 120 240 480; do
for cfg in "VGMI_HMM_WAVES=2" "VGMI_HMM_WAVES=4"; do
  echo "windows $w $cfg: $(env VGMI_HMM_TIMING=1 $cfg python3 tools/bench_hmm.py 1000 $w 2>&1 | grep -o 'recursion [0-9.]* ms' | sort -k2 -n | head -1)"
done; done
