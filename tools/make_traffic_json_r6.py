#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_<w>/traffic.json (copied to profiles/hbm_traffic[_c3|_c5].json by hand once judged) from the summary
tools/profile_r6.sh writes.

MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE come from their own --pmc passes, in KiB.  FETCH_SIZE tallies 64 B per
128 B request on wide coalesced streams: the row loads of the count kernels are non-temporal loads of whole rows, calibrated in round 1
on the C2 kernel's own stream (factor 2.0037: the same launch with the candidate runs dropped read exactly the read block), so the known
stream bytes are counted in full and what FETCH_SIZE shows beyond stream / factor -- buckets, their triples, path-table lines, the
overflow table -- is taken raw, as is WRITE_SIZE (counter atomics, queued run records).  Every kernel of the count pass is summed
(`kernels`: name -> calls, average us, counters); `kernel` is the dominant one.  The file carries the sha256 of the libvgmi.so the
passes ran: bench.py reports it as `roofline.traffic` only for that library and that kernel."""
import json
import os
import re
import sys

CAL = 2.0036931669084184      # profiles/r1: FETCH_SIZE calibration on the same row loads


def main():
    w, n_reads, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rx = re.compile(sys.argv[4] if len(sys.argv) > 4 else r"count27|countkc|ctd_")
    known_stream = w != "bloom"      # the read block of the count kernels: the calibrated 12- / 16-byte-per-lane row loads (K3: no calibrated stream, everything raw)
    pmc, trace = {}, {}
    for ln in open(f"{out}/summary.txt"):
        m = re.match(r"\s*(\d+)\s+[\d.]+\s+([\d.]+)\s+([A-Z]\S+)\s+(.*)$", ln)          # dispatches sum per_dispatch counter kernel
        if m and rx.search(m.group(4)):
            pmc.setdefault(m.group(4).strip(), {})[m.group(3)] = (int(m.group(1)), float(m.group(2)))
            continue
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+(.*)$", ln)          # calls total avg pct name
        if m and rx.search(m.group(4)):
            trace[m.group(4).strip()] = (int(m.group(1)), float(m.group(3)), float(m.group(2)))
    if not trace:
        sys.exit("no count kernel in the kernel table")

    def short(n):
        return n.split("(")[0].replace("void ", "").strip()
    # the launches that process the whole block: those of the dominant kernel; per-launch figures = sums over the pass's kernels / that count
    dom = max(trace, key=lambda n: trace[n][2])
    steps_trace = trace[dom][0]
    kernels = {}
    tot = {}
    for name, cs in pmc.items():
        for c, (nd, per) in cs.items():
            dom_nd = max((v[c][0] for k, v in pmc.items() if short(k) == short(dom) and c in v), default=nd)
            tot[c] = tot.get(c, 0.0) + per * nd / dom_nd
        kernels[short(name)[:80]] = {c: per for c, (nd, per) in cs.items()}
    pass_us = sum(t[2] for t in trace.values()) / steps_trace
    stream = n_reads * 151 if known_stream else 0
    fetch, write = tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
    under = stream * (1 - 1 / CAL)
    d = {"reads_per_launch": n_reads, "kernel": short(dom) if known_stream else None, "libvgmi_sha256": open(f"{out}/libvgmi.sha256").read().strip(),
         "source_sha256": open(f"{out}/source.sha256").read().strip() if os.path.exists(f"{out}/source.sha256") else None,
         "fetch_size_kb_raw": tot["FETCH_SIZE"], "write_size_kb_raw": tot["WRITE_SIZE"], "stream_bytes_known": stream,
         "fetch_size_calibration_factor_streaming": CAL,
         "kernel_avg_us_rocprofv3": trace[dom][1], "kernel_calls_rocprofv3": trace[dom][0], "count_pass_us_per_launch_rocprofv3": pass_us,
         "kernel_table": {short(n)[:80]: {"calls": t[0], "avg_us": t[1]} for n, t in trace.items()},
         "bytes_per_launch": int(fetch + under + write),
         "breakdown_bytes": {"fetch_raw": int(fetch), "stream_undercount_added": int(under), "write_raw": int(write)},
         "method": __doc__.split("\n\n", 1)[1].replace("\n", " ")}
    if "TCC_EA0_RDREQ_sum" in tot:
        d.update({"ea_read_requests": int(tot["TCC_EA0_RDREQ_sum"]), "ea_write_requests": int(tot["TCC_EA0_WRREQ_sum"]),
                  "ea_write_requests_64B": int(tot["TCC_EA0_WRREQ_64B_sum"]), "l2_hits": int(tot["TCC_HIT_sum"]), "l2_misses": int(tot["TCC_MISS_sum"]),
                  "tcp_tcc_read_requests": int(tot["TCP_TCC_READ_REQ_sum"]),
                  "tcp_tcc_atomic_requests_without_return": int(tot["TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"]),
                  "tcp_tcc_atomic_requests_with_return": int(tot["TCP_TCC_ATOMIC_WITH_RET_REQ_sum"]),
                  "memory_side_requests_per_read": (tot["TCC_EA0_RDREQ_sum"] + tot["TCC_EA0_WRREQ_sum"]) / n_reads,
                  "sq": {k: v for k, v in tot.items() if k.startswith("SQ_")}})
    d["per_kernel_counters"] = kernels
    json.dump(d, open(f"{out}/traffic.json", "w"), indent=1)
    print(w, d["kernel"], round(d["bytes_per_launch"] / 1e9, 2), "GB per launch;", round(d.get("memory_side_requests_per_read", 0), 2),
          "requests per read; dominant kernel", trace[dom][1], "us; pass", round(pass_us, 1), "us")


if __name__ == "__main__":
    main()
