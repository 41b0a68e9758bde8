#!/usr/bin/env python3
"""profiles/hbm_traffic_c3.json / hbm_traffic_c5.json from the summaries tools/profile_r4.sh writes (gpurun_out/prof_r4_<w>/summary.txt).

MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE come from their own --pmc passes, in KiB.  FETCH_SIZE tallies 64 B per
128 B request on wide coalesced streams: the row loads of count27c_kernel are the 12-byte-per-lane non-temporal loads of whole 768-byte
rows the C2 kernel's stream was calibrated on (profiles/hbm_traffic.json, factor 2.0037), so the known stream bytes are counted in
full and what FETCH_SIZE shows beyond half of them -- buckets, their triples, the overflow table -- is taken raw, as is WRITE_SIZE (the
counter atomics).  The request counters of the same launches are listed because the kernel is bound by memory-side requests, not bytes."""
import json
import re
import sys

CAL = 2.0036931669084184      # profiles/hbm_traffic.json: FETCH_SIZE calibration on the same row loads


def main():
    w = sys.argv[1]
    n_reads = int(sys.argv[2])
    summ = f"gpurun_out/prof_r4_{w}/summary.txt"
    vals, avg_us, calls = {}, None, None
    for ln in open(summ):
        m = re.match(r"\s*(\d+)\s+[\d.]+\s+([\d.]+)\s+(\S+)\s+.*count27c_kernel", ln)
        if m:
            vals[m.group(3)] = float(m.group(2))
        m = re.match(r"\s*(\d+)\s+[\d.]+\s+([\d.]+)\s+[\d.]+\s+vgk::count27c_kernel", ln)
        if m:
            calls, avg_us = int(m.group(1)), float(m.group(2))
    stream = n_reads * 151
    fetch, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    under = stream * (1 - 1 / CAL)
    d = {"reads_per_launch": n_reads, "kernel": "vgk::count27c_kernel", "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb_raw": vals["WRITE_SIZE"],
         "stream_bytes_known": stream, "fetch_size_calibration_factor_streaming": CAL,
         "ea_read_requests": int(vals["TCC_EA0_RDREQ_sum"]), "ea_write_requests": int(vals["TCC_EA0_WRREQ_sum"]),
         "ea_write_requests_64B": int(vals["TCC_EA0_WRREQ_64B_sum"]), "l2_hits": int(vals["TCC_HIT_sum"]), "l2_misses": int(vals["TCC_MISS_sum"]),
         "tcp_tcc_read_requests": int(vals["TCP_TCC_READ_REQ_sum"]),
         "tcp_tcc_atomic_requests_without_return": int(vals["TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"]),
         "memory_side_requests_per_read": (vals["TCC_EA0_RDREQ_sum"] + vals["TCC_EA0_WRREQ_sum"]) / n_reads,
         "kernel_avg_us_rocprofv3": avg_us, "kernel_calls_rocprofv3": calls,
         "sq": {k: vals[k] for k in vals if k.startswith("SQ_")},
         "method": __doc__.split("\n\n", 1)[1].replace("\n", " "),
         "bytes_per_launch": int(fetch + under + write),
         "breakdown_bytes": {"fetch_raw": int(fetch), "stream_undercount_added": int(under), "write_raw": int(write)},
         "summary": f"profiles/r4_{w}_rocprofv3_summary.txt"}
    json.dump(d, open(f"profiles/hbm_traffic_{w}.json", "w"), indent=1)
    print(w, d["bytes_per_launch"] / 1e9, "GB;", round(d["memory_side_requests_per_read"], 2), "requests per read;", avg_us, "us")


if __name__ == "__main__":
    main()
