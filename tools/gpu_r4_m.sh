#!/bin/bash
# block-gzip inflate: round-4 batch kernel against round 3's (same box), then the ingest tests on the new one
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4m; mkdir -p $O
timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 256 > $O/new.json 2> $O/new.err; tail -1 $O/new.json; tail -2 $O/new.err
cp varigraph_amd/csrc/vgmi_inflate.hip $O/new_inflate.hip
cp tools/ab/inflate_r3.hip.txt varigraph_amd/csrc/vgmi_inflate.hip
python3 -m varigraph_amd.build > /dev/null 2>&1
timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 256 > $O/old.json 2> $O/old.err; tail -1 $O/old.json
cp $O/new_inflate.hip varigraph_amd/csrc/vgmi_inflate.hip
python3 -m varigraph_amd.build > /dev/null 2>&1
timeout 600 python3 tools/bench_bgzf_only.py 8000000 4 256 > $O/new2.json 2> $O/new2.err; tail -1 $O/new2.json
timeout 900 python -m pytest tests/test_gpu_ingest.py -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
