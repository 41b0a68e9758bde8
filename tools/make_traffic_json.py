#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the summary tools/profile_r1.sh writes (gpurun_out/prof_r1/summary.txt).

MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE come from their own --pmc passes, in KiB.  FETCH_SIZE tallies
64 B per 128 B request on wide coalesced streams; the factor is measured on this kernel's own row loads (the
calibration pass: same launch with the candidate runs dropped, VGMI_DBG=1 in an ablation build, which reads exactly the read
block plus the filter staging) and applied to the streaming part only.  The remainder -- index buckets, unitig sequence and
bit words (round 3; table probes before) that miss the XCD L2 -- and WRITE_SIZE are taken raw."""
import json, re, sys, os

def main():
    summ = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r1/summary.txt"
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/hbm_traffic.json"
    n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
    read_len = 150
    sect = None
    vals = {}
    kernel = None
    for ln in open(summ):
        m = re.match(r"## PMC: (\S+)", ln)
        if m:
            sect = "cal" if "pmc_cal" in m.group(1) else "main"
            continue
        m = re.match(r"\s*\d+\s+[\d.]+\s+([\d.]+)\s+(FETCH_SIZE|WRITE_SIZE)\s+(.*count27[xs]?_kernel\S*)", ln)
        if m and sect:
            vals[(sect, m.group(2))] = float(m.group(1))
            kernel = m.group(3).split("(")[0].replace("void ", "").strip()
    fetch, cal, write = vals[("main", "FETCH_SIZE")], vals[("cal", "FETCH_SIZE")], vals[("main", "WRITE_SIZE")]
    stream = n_reads * (read_len + 1) + 256 * (128 << 10)    # the read block once + one 128 KiB filter staging per workgroup (256 CUs)
    factor = stream / (cal * 1024)
    probe = (fetch - cal) * 1024
    wr = write * 1024
    d = {
        "reads_per_launch": n_reads,
        "kernel": kernel,
        "fetch_size_kb_raw": fetch,
        "fetch_size_kb_stream_only_raw": cal,
        "write_size_kb_raw": write,
        "stream_bytes_known": stream,
        "fetch_size_calibration_factor_streaming": factor,
        "method": __doc__.split("\n\n", 1)[1].replace("\n", " "),
        "bytes_per_launch": int(stream + probe + wr),
        "breakdown_bytes": {"read_stream": int(stream), "table_probe_l2_misses": int(probe), "writes_atomics": int(wr)},
    }
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d["breakdown_bytes"]), "factor", round(factor, 4))

if __name__ == "__main__":
    main()
