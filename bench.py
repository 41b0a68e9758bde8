#!/usr/bin/env python3
"""bench.py -- reads/s of the per-sample genotyping hot path on MI355X (BASELINE.json metric).

One "step" = one whole sample pass of the hot path over one read block that is already resident
in HBM: reset counters -> K1/K2 read counting (the dominant kernel) -> K5/K6 clamp + per-node
gather + coverage histogram, results left on the device.

Workload at N=1 (BASELINE.json configs[1], "C2"): the 1 Mb / 1 k SNP / 15-haplotype graph
(tests/golden/c1/graph.bin.gz, built by the real reference `construct`), one sample of
50 M 2x150 bp read pairs = 1e8 reads, k = 27, generated on the device by the seeded generator
(varigraph_amd/csrc/bench/vg_synth.h: libvgsynth.so, bench / test tooling).  With --gpus N every rank processes its own sample (seed
1000+rank) after ONE RCCL broadcast of the table image from rank 0: weak scaling, no data-path
collective.

Next to the C2 numbers the same line carries a `c3` block (BASELINE.json configs[2] / [3]): the chr20-class graph
(60 Mb reference, 500 k SNPs, 2.56e7 k-mers: the table lives in HBM) with 2.4e7 device-generated reads per rank, its
own kernel time, measured hits per read and HBM roofline (B_read = 150 + 992 + 2 hits), and a `verify`
block: an UNSATURATED prefix of each workload compared counter by counter with the oracle outside the timed region.

`python bench.py --gpus N` without a launcher starts its own N ranks (one process per GPU, RCCL); under
`torch.distributed.run` it uses the ranks it is given.

Prints one JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
K = 27
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def large_kernel_name():
    """the count kernel of graphs that live in HBM (k = 27, > 65 536 k-mers), as the environment's A/B knobs select it"""
    if os.environ.get("VGMI_XTABLE") == "0":
        return "vgk::count27_kernel<false, true>"
    return "vgk::count27x_kernel" if os.environ.get("VGMI_CTABLE") == "0" else "vgk::count27c_kernel"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_graph():
    """graph.bin of the C1/C2 configuration -> keys, node CSR, hom flags (product loader)."""
    from varigraph_amd import host
    g = host.load_graph(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"))
    return g


def cohort_haplotypes():
    from varigraph_amd import synth
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "c1", "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    variants, gts = synth.make_cohort(ref, meta["n_var"], n_samples=meta["n_samples"], ploidy=meta["ploidy"],
                                      seed=meta["cohort_seed"], indel_frac=meta["indel_frac"], sv_frac=meta["sv_frac"])
    return synth.sample_haplotypes(ref, variants, gts, 0, meta["ploidy"])


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_quota():
    """CPUs this process may use at once according to its cgroup (cpu.max), or None without a quota."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(haps, n_reads, cores):
    """Reference CPU path (oracle/_ref/ref_harness = the unmodified reference's
    FastqKmer::build_fastq_index) on a bounded sample of the same workload, on this box's host
    cores.  The reference parses on its main thread, so more threads stop helping early; a short
    sweep picks the best setting and `cores` reports the thread count of the reported value."""
    from varigraph_amd import synth, vgmi
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    graph_gz = os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz")
    work = tempfile.mkdtemp(prefix="vg_cpu_")
    try:
        block = vgmi.synth_reads_host(1000, 0, n_reads, READ_LEN, haps)
        if os.path.exists(harness):
            import gzip
            graph = os.path.join(work, "graph.bin")
            with open(graph, "wb") as f:
                f.write(gzip.open(graph_gz, "rb").read())
            fq = synth.write_fastq_pair_fast(os.path.join(work, "s"), block, n_reads, READ_LEN)
            sweep = {}
            t_budget = time.perf_counter() + 40.0
            for threads in sorted({10, 16, 32, min(64, cores), cores}):  # 10 = reference default (-t)
                if threads > cores or time.perf_counter() > t_budget:
                    continue
                try:
                    # the reference's own thread pool notifies without holding its mutex (include/ThreadPool.hpp:
                    # submit / shutdown), so a run can -- rarely -- sleep forever on a lost wake-up: bound it
                    out = subprocess.run([harness, "count", graph, str(threads), os.path.join(work, "c.bin")] + fq,
                                         capture_output=True, text=True, timeout=120)
                except subprocess.TimeoutExpired:
                    log(f"reference harness at -t {threads} did not finish in 120 s (lost wake-up in its thread pool?): skipped")
                    continue
                if out.returncode != 0:
                    log("reference harness failed:", out.stderr[-500:])
                    continue
                vals = dict(ln.split(" ", 1) for ln in out.stdout.splitlines() if " " in ln)
                sweep[threads] = n_reads / float(vals["build_fastq_index_s"])
            t1 = None
            if sweep and time.perf_counter() < t_budget:
                # -t 1 on the first tenth of the sample (the single-thread rate is ~1e5 reads/s)
                n1 = (n_reads // 10) // 2 * 2
                fq1 = synth.write_fastq_pair_fast(os.path.join(work, "s1"), block[: n1 * (READ_LEN + 1)], n1, READ_LEN)
                try:
                    out = subprocess.run([harness, "count", graph, "1", os.path.join(work, "c1.bin")] + fq1,
                                         capture_output=True, text=True, timeout=120)
                    if out.returncode == 0:
                        vals = dict(ln.split(" ", 1) for ln in out.stdout.splitlines() if " " in ln)
                        t1 = round(n1 / float(vals["build_fastq_index_s"]))
                except subprocess.TimeoutExpired:
                    pass
            if sweep:
                best = max(sweep, key=sweep.get)
                return {"value": sweep[best], "unit": "reads/s", "cores": best, "kind": "reference",
                        "host_logical_cpus": cores, "cgroup_cpu_quota": cpu_quota(), "cpu_model": cpu_model(),
                        "reads_per_s_at_t1": t1,
                        "sweep_reads_per_s_by_threads": {str(k): round(v) for k, v in sorted(sweep.items())},
                        "sample": f"{n_reads} reads (plain FASTQ, 2 files) of the same workload through the unmodified "
                                  f"reference FastqKmer::build_fastq_index (oracle/_ref), best of the -t sweep"}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        from varigraph_amd import host
        g = host.load_graph(graph_gz)
        t = oracle_lib.Table(g["keys"])
        t0 = time.perf_counter()
        t.count_block(block, K)
        secs = time.perf_counter() - t0
        return {"value": n_reads / secs, "unit": "reads/s", "cores": 1, "kind": "port",
                "sample": f"{n_reads} reads of the same workload (in-memory block), oracle/vg_oracle.c, {secs:.2f} s"}
    finally:
        import shutil
        shutil.rmtree(work, ignore_errors=True)


def sample_level(ctx, haps, cpu_ref, n_plain=8_000_000, n_packed=8_000_000):
    """SURVEY 8d level (ii): FASTQ FILES -> counters, through the product's FastqKmerHip (csrc/host): plain, gzip and
    block-gzip copies of one sample of the C2 workload, with the records found on the device (vgmi_fastq_*) and, for
    comparison, by the host parser; plus the PCIe-inclusive rate of the host-block entry point vgmi_reads_submit."""
    import gzip
    import shutil
    from varigraph_amd import host, synth, vgmi
    out = {}
    work = tempfile.mkdtemp(prefix="vg_sample_")
    try:
        g = host.Graph(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"))
        g.upload(ctx)
        threads = int(min(16, max(2, (cpu_quota() or os.cpu_count() or 2))))
        block = vgmi.synth_reads_host(1000, 0, n_plain, READ_LEN, haps)
        plain = synth.write_fastq_pair_fast(os.path.join(work, "p"), block, n_plain, READ_LEN)
        small = synth.write_fastq_pair_fast(os.path.join(work, "c"), block[: n_packed * (READ_LEN + 1)], n_packed, READ_LEN)
        def to_gzip(p):
            with open(p, "rb") as fi, gzip.open(p + ".gz", "wb", compresslevel=4) as fo:
                shutil.copyfileobj(fi, fo, 1 << 24)
            return p + ".gz"

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(4) as pool:   # zlib releases the interpreter lock: the four files are written side by side
            gz_f = [pool.submit(to_gzip, p) for p in small]
            bgz_f = [pool.submit(synth.bgzf_compress_file, p, p + ".bgz.gz", 4) for p in small]
            gz, bgz = [f.result() for f in gz_f], [f.result() for f in bgz_f]
        out["bytes_per_read_plain"] = sum(os.path.getsize(p) for p in plain) / n_plain
        out["host_threads"] = threads

        def rate(files, n, host_parse, n_threads=None):
            os.environ["VGH_HOST_PARSE"] = "1" if host_parse else "0"
            best, cov = None, None
            for _ in range(3):
                t0 = time.perf_counter()
                cov, _, _, st = g.sample_count(ctx, files, threads=n_threads or threads, require_depth=False)
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            return n / best, cov

        r_plain, cov_ref = rate(plain, n_plain, False)
        out["plain_reads_per_s"] = r_plain
        out["plain_text_gb_per_s"] = r_plain * out["bytes_per_read_plain"] / 1e9
        out["plain_host_parser_reads_per_s"], cov_h = rate(plain, n_plain, True)
        # ordinary gzip: inflated on the device since round 4 (vgmi_gunzip.hip) -- the host threads only read the file; the rate with
        # four threads in all (two per stream) is the one that matters for a node whose eight GPUs share the host, and
        # VGH_DEVICE_GUNZIP=0 is rounds 2-3's path (several host inflate threads per stream) on the same files
        out["gzip_reads_per_s"], cov_gz = rate(gz, n_packed, False)
        out["gzip_reads_per_s_4_host_threads"], _ = rate(gz, n_packed, False, 4)
        os.environ["VGH_DEVICE_GUNZIP"] = "0"
        out["gzip_host_inflate_reads_per_s"], cov_gz_h = rate(gz, n_packed, False)
        out["gzip_host_inflate_reads_per_s_4_host_threads"], _ = rate(gz, n_packed, False, 4)
        os.environ.pop("VGH_DEVICE_GUNZIP", None)
        out["gzip_counters_identical_device_vs_host_inflate"] = bool(np.array_equal(cov_gz, cov_gz_h))
        out["bgzf_reads_per_s"], _ = rate(bgz, n_packed, False)
        out["bgzf_reads_per_s_4_host_threads"], _ = rate(bgz, n_packed, False, 4)
        # the same reads with four quality values drawn at random (what a binned instrument writes: the files above carry one constant
        # quality, SURVEY 8d's specification, and compress 1.65 x better than these -- the compressed rates on them are the kind ones)
        qdir = os.path.join(work, "q")
        os.makedirs(qdir)
        binned = synth.write_fastq_pair_fast(os.path.join(qdir, "b"), block[: n_packed * (READ_LEN + 1)], n_packed, READ_LEN, qual="binned")
        with ThreadPoolExecutor(4) as pool:
            gz_f = [pool.submit(to_gzip, p) for p in binned]
            bgz_f = [pool.submit(synth.bgzf_compress_file, p, p + ".bgz.gz", 4) for p in binned]
            gz_b, bgz_b = [f.result() for f in gz_f], [f.result() for f in bgz_f]
        out["compressed_bytes_per_read"] = {"gzip_constant_quality": sum(os.path.getsize(p) for p in gz) / n_packed,
                                            "gzip_binned_qualities": sum(os.path.getsize(p) for p in gz_b) / n_packed,
                                            "bgzf_constant_quality": sum(os.path.getsize(p) for p in bgz) / n_packed,
                                            "bgzf_binned_qualities": sum(os.path.getsize(p) for p in bgz_b) / n_packed}
        out["gzip_reads_per_s_binned_qualities"], cov_gz_b = rate(gz_b, n_packed, False)
        out["bgzf_reads_per_s_binned_qualities"], cov_bgz_b = rate(bgz_b, n_packed, False)
        out["binned_qualities_counters_identical"] = bool(np.array_equal(cov_gz_b, cov_gz) and np.array_equal(cov_bgz_b, cov_gz))
        out["counters_identical_device_vs_host_parser"] = bool(np.array_equal(cov_ref, cov_h))
        os.environ.pop("VGH_HOST_PARSE", None)
        # PCIe-inclusive: the packed read block handed over from host memory (vgmi_reads_submit: pinned staging + H2D)
        best = None
        for _ in range(3):
            ctx.counts_reset()
            t0 = time.perf_counter()
            ctx.reads_submit(block, n_plain)
            ctx.counts_finish()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        out["reads_submit_from_host_memory_reads_per_s"] = n_plain / best
        if cpu_ref and cpu_ref.get("kind") == "reference":
            out["reference_plain_reads_per_s"] = cpu_ref["value"]
            out["plain_vs_reference"] = r_plain / cpu_ref["value"]
        harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
        if os.path.exists(harness):   # the unmodified reference on the gzip files, -t 10 (its default)
            graph = os.path.join(work, "graph.bin")
            with open(graph, "wb") as f:
                f.write(gzip.open(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"), "rb").read())
            try:
                r = subprocess.run([harness, "count", graph, "10", os.path.join(work, "c.bin")] + gz, capture_output=True, text=True,
                                   timeout=120)
                vals = dict(ln.split(" ", 1) for ln in r.stdout.splitlines() if " " in ln)
                out["reference_gzip_reads_per_s"] = n_packed / float(vals["build_fastq_index_s"])
                out["gzip_vs_reference"] = out["gzip_reads_per_s"] / out["reference_gzip_reads_per_s"]
            except (subprocess.TimeoutExpired, KeyError, ValueError):
                pass
        out["sample"] = (f"{n_plain} reads (2 plain FASTQ files, {out['bytes_per_read_plain']:.0f} B/read) / {n_packed} reads "
                         f"(gzip and block-gzip level 4), page-cache resident, best of 3; files -> counters on the host")
        g.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def c4_cli(n_samples, genome, variants, pairs, threads, repeats=2):
    """BASELINE.json configs[3] at the level a user runs it: `varigraph-mi genotype` on n_samples chr20-scale samples in one `-s`
    list over every device present (one device: its contexts share it), from plain FASTQ files, wall clock of the whole command --
    graph load, counting, HMM, VCFs -- and the stages its own log reports (VGH_TIMING=1).  The counting kernels are a few per cent
    of this; what it measures is what a node of eight GPUs would be bound by (the host's share per sample)."""
    import re
    import shutil
    import subprocess
    import tempfile
    import torch
    from varigraph_amd import synth
    cli = os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi")
    if not os.path.exists(cli):
        return {"error": "varigraph_amd/bin/varigraph-mi is not built"}
    work = tempfile.mkdtemp(prefix="vg_c4_")
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022", VGH_TIMING="1")
    out = {"workload": f"C4 through the CLI: {n_samples} samples x {pairs} read pairs 2x150 bp ({2 * pairs * READ_LEN / genome:.0f}x) over a "
                       f"{genome // 1_000_000} Mb / {variants} variant / 15-haplotype graph, `varigraph-mi genotype -t {threads}`, plain FASTQ",
           "samples": n_samples, "threads": threads}
    try:
        t0 = time.perf_counter()
        ref = synth.make_reference(genome)
        var, gts = synth.make_cohort(ref, variants, n_samples=7, ploidy=2, seed=11)
        fa, vcf = os.path.join(work, "ref.fa"), os.path.join(work, "in.vcf")
        synth.write_fasta(fa, "chr1", ref)
        synth.write_vcf(vcf, "chr1", len(ref), var, gts, 7, 2)
        haps = synth.sample_haplotypes(ref, var, gts, 0, 2)
        fq = synth.write_fastq_pair_device(os.path.join(work, "s"), haps, pairs, 1000)
        out["files_s"] = time.perf_counter() - t0
        graph = os.path.join(work, "graph.bin")
        t0 = time.perf_counter()
        r = subprocess.run([cli, "construct", "-r", fa, "-v", vcf, "--save-graph", graph, "-t", "32", "--gpu", "0"], cwd=work,
                           capture_output=True, text=True, env=env)
        out["construct_s"] = time.perf_counter() - t0
        if r.returncode != 0:
            out["error"] = r.stderr[-400:]
            return out
        n_dev = max(1, torch.cuda.device_count())
        gpus = ",".join(str(i) for i in range(n_dev))
        out["gpus"] = gpus
        d = os.path.join(work, "run")
        best = None
        for _ in range(repeats):
            shutil.rmtree(d, ignore_errors=True)
            os.makedirs(d)
            open(os.path.join(d, "samples.cfg"), "w").write("".join(f"sample{i} " + " ".join(fq) + "\n" for i in range(n_samples)))
            t0 = time.perf_counter()
            r = subprocess.run([cli, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", str(threads), "--gpus", gpus], cwd=d,
                               capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                out["error"] = r.stderr[-400:]
                return out
            log = r.stderr

            def nums(pat):
                return [float(x) for x in re.findall(pat, log)]
            # (the four passes around the device calls, each measured around its whole body: the hidden-state / emission / posterior timers of
            # the all-host path tick INSIDE passes A and C when the emissions run on the device -- they are listed, not added)
            host = {"node_lists": sum(nums(r"node lists ([\d.]+),")), "host_scored_nodes_and_genotype_strings": sum(nums(r"genotype strings ([\d.]+),")),
                    "step_tables": sum(nums(r"step tables ([\d.]+),")), "calls_and_vcf_lines": sum(nums(r"calls \+ VCF lines ([\d.]+),")),
                    "coverage_words": sum(nums(r"coverage words ([\d.]+),")), "text_joined": sum(nums(r"text joined ([\d.]+),"))}
            inside = {"hidden_states": sum(nums(r"hidden states ([\d.]+),")), "emissions": sum(nums(r"emissions ([\d.]+), forward")),
                      "posterior": sum(nums(r"posterior ([\d.]+) \(")), "selection": sum(nums(r"selection ([\d.]+),"))}
            run = {"genotype_wall_s": dt, "samples_per_s": n_samples / dt,
                   "graph_load_s": (nums(r"graph loaded: .*\(([\d.]+) s\)") or [None])[0],
                   "counting_wall_s_per_sample": float(np.mean(nums(r"counting ([\d.]+) s"))) if nums(r"counting ([\d.]+) s") else None,
                   "count_kernel_s_per_sample": float(np.mean(nums(r"\(kernel ([\d.]+) s"))) if nums(r"\(kernel ([\d.]+) s") else None,
                   "hmm_device_recursion_s_per_sample": float(np.mean(nums(r"HMM recursion on the device: ([\d.]+) s"))) if nums(r"HMM recursion on the device: ([\d.]+) s") else None,
                   "genotyping_wall_s_per_sample": float(np.mean(nums(r"genotyping ([\d.]+) s"))) if nums(r"genotyping ([\d.]+) s") else None,
                   "vcf_text_and_gzip_s_per_sample": float(np.mean([a + b for a, b in zip(nums(r"VCF text ([\d.]+),"), nums(r"gzip ([\d.]+)\)"))])) if nums(r"VCF text ([\d.]+),") else None,
                   "host_thread_seconds": host, "host_thread_seconds_per_sample": sum(host.values()) / n_samples,
                   "of_which_inside_the_passes": inside}
            run["_log"] = log
            if os.environ.get("VG_BENCH_C4_LOG"):
                run["log"] = [ln for ln in log.split("\n") if any(k in ln for k in ("HMM part", "thread-seconds", "counting ", "genotyping ", "graph loaded", "consumer", "recursion", "done in"))]
            if best is None or dt < best["genotype_wall_s"]:
                best = run
        out.update(best)
        # one PROCESS per device (--procs; graph.bin parsed once before the fork, the ranks share it copy-on-write): its wall next to the
        # one-process run's above, and the host memory both report (proportional set size at exit, peak resident size)
        def mem(log):
            rows = [ln for ln in log.split("\n") if "host memory: peak RSS" in ln]
            return {"processes": len(rows), "peak_rss_gb_summed": sum(float(ln.split("peak RSS ")[1].split(" GB")[0]) for ln in rows),
                    "pss_at_exit_gb_summed": sum(float(ln.split("PSS at exit ")[1].split(" GB")[0]) for ln in rows)}
        out["host_memory"] = mem(best.get("_log", ""))
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        open(os.path.join(d, "samples.cfg"), "w").write("".join(f"sample{i} " + " ".join(fq) + "\n" for i in range(n_samples)))
        t0 = time.perf_counter()
        r = subprocess.run([cli, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", str(threads), "--gpus", gpus, "--procs"], cwd=d,
                           capture_output=True, text=True, env=env)
        out["procs"] = {"genotype_wall_s": time.perf_counter() - t0, "returncode": r.returncode, "host_memory": mem(r.stderr),
                        "rccl_broadcast": "RCCL broadcast" in r.stderr,
                        "note": "`--procs --gpus " + gpus + "`: one rank per device named; a single rank has nobody to send to and makes no communicator"}
        out.pop("_log", None)
        out["note"] = ("best of %d runs; the %d samples read the same two FASTQ files (page cache); eight devices would each run one sample's "
                       "counting + device HMM side by side, the host thread-seconds per sample are what they share" % (repeats, n_samples))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def measured_traffic(which, n_reads, kernel):
    """HBM-side bytes per launch of a count pass from the committed rocprofv3 PMC passes (profiles/hbm_traffic[_c3|_c5|_bloom].json,
    written by tools/profile_r6.sh + tools/make_traffic_json_r6.py) -- ONLY when the file was measured on this code (sha256 of the
    library or of its sources), on this kernel and on a launch of this size; anything else is a claim about another build: None."""
    from varigraph_amd import build
    p = os.path.join(ROOT, "profiles", {"c2": "hbm_traffic.json"}.get(which, f"hbm_traffic_{which}.json"))
    if not os.path.exists(p):
        return None, None
    t = json.load(open(p))
    if t.get("reads_per_launch") != n_reads or (kernel and t.get("kernel") != kernel):
        return None, None
    if t.get("libvgmi_sha256") != build.lib_digest() and t.get("source_sha256") != build.source_digest():
        return None, None
    return t["bytes_per_launch"], {k: t.get(k) for k in ("kernel", "libvgmi_sha256", "source_sha256", "fetch_size_kb_raw", "write_size_kb_raw",
                                                        "breakdown_bytes", "memory_side_requests_per_read", "count_pass_us_per_launch_rocprofv3")}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, BEFORE anything in
    this process touches the GPU (the parent never does), and leave with the worst exit code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed sample passes (C2: 6 ms each)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per sample (2 per pair)")
    ap.add_argument("--cpu-reads", type=int, default=4_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo + VGMI_BENCH_DEVICE=0 lets several ranks share one GPU "
                         "to exercise the N>1 control flow on a single-GPU box (debug only)")
    ap.add_argument("--shard-reads", action="store_true",
                    help="strong scaling: ONE sample of --reads reads sharded over the ranks, raw counters summed "
                         "with one RCCL all-reduce per step (default: one sample per rank, weak scaling)")
    ap.add_argument("--no-c3", action="store_true", help="skip the chr20-class (table in HBM) leg")
    ap.add_argument("--no-other-k", action="store_true", help="skip the k = 25 / 21 / 22 / 28 sub-block of the chr20-class leg")
    ap.add_argument("--c3-reads", type=int, default=24_000_000)
    ap.add_argument("--c3-steps", type=int, default=20)
    ap.add_argument("--c3-genome", type=int, default=60_000_000, help="(reduced sizes: multi-rank recipes on one GPU)")
    ap.add_argument("--c3-variants", type=int, default=500_000)
    ap.add_argument("--c3-contexts", type=int, default=5, help="fresh contexts the chr20-class kernel time is taken over (median, min, max)")
    ap.add_argument("--no-c5", action="store_true", help="skip the whole-genome-class (3 Gb, 5 M SNPs) leg")
    ap.add_argument("--no-bloom", action="store_true", help="skip the construct-side Bloom leg")
    ap.add_argument("--c5-reads", type=int, default=100_000_000)
    ap.add_argument("--c5-steps", type=int, default=10)
    ap.add_argument("--c5-genome", type=int, default=3_000_000_000)
    ap.add_argument("--c5-variants", type=int, default=5_000_000)
    ap.add_argument("--c5-contexts", type=int, default=3)
    ap.add_argument("--verify-reads", type=int, default=1_000_000, help="unsaturated prefix checked against the oracle")
    ap.add_argument("--unsaturated-reads", type=int, default=3_000_000, help="C2: the first launch of this many reads after a reset (roofline.unsaturated)")
    ap.add_argument("--no-sample-level", action="store_true", help="skip the FASTQ-files-to-counters leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the CLI-level eight-sample leg (BASELINE configs[3] through varigraph-mi)")
    ap.add_argument("--c4-samples", type=int, default=8)
    ap.add_argument("--c4-pairs", type=int, default=12_000_000, help="read pairs per sample (SURVEY 8: 30x of 60 Mb = 12 M pairs)")
    ap.add_argument("--c4-threads", type=int, default=10)
    ap.add_argument("--budget-s", type=float, default=480.0,
                    help="wall clock for the whole command: a leg whose estimate does not fit what is left is skipped ({'skipped': 'budget'}), the headline never")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    t_start = time.perf_counter()

    import torch
    from varigraph_amd import build, vgmi
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if "VGMI_BENCH_DEVICE" in os.environ:
        local = int(os.environ["VGMI_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    comm_dev = torch.device("cuda", local) if args.backend == "nccl" else torch.device("cpu")
    ctx_dev = torch.device("cuda", local)
    t_init = None
    if world > 1:
        import torch.distributed as dist
        t0 = time.perf_counter()
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=ctx_dev)
        else:
            dist.init_process_group("gloo")
        t_init = time.perf_counter() - t0
    if rank == 0:
        build.build_vgmi()
        build.build_host()
    if dist:
        dist.barrier()

    from varigraph_amd import dist as vdist
    ctx = vgmi.Context(local, buffer_mib=256)
    shard = args.shard_reads and world > 1

    # ---- the line: built up leg by leg; printed ONCE, at the end -- or by the handler below if the command is told to stop earlier
    out = {}
    printed = [False]

    def emit():
        if rank == 0 and out and not printed[0]:
            printed[0] = True
            print(json.dumps(out), flush=True)

    if rank == 0:
        import atexit
        import signal

        def on_term(signum, frame):
            out.setdefault("interrupted", f"signal {signum} after {time.perf_counter() - t_start:.0f} s: the legs still to come are absent")
            emit()
            os._exit(0 if "value" in out else 1)
        signal.signal(signal.SIGTERM, on_term)
        atexit.register(emit)

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def agree_max(x):
        return vdist.max_over_ranks(float(x), dist, comm_dev) if dist else float(x)

    def run_leg(name, estimate_s, fn, where=None):
        """One optional leg: skipped when its estimate does not fit the budget (the same decision on every rank), a failure inside it is
        recorded in its block instead of costing the line."""
        where = out if where is None else where
        used = agree_max(time.perf_counter() - t_start)
        if used + estimate_s > args.budget_s:
            if rank == 0:
                where[name] = {"skipped": "budget", "seconds_used": round(used, 1), "estimate_s": estimate_s, "budget_s": args.budget_s}
                log(f"[bench] {name}: skipped (budget)")
            return None
        t0 = time.perf_counter()
        try:
            res = fn()
        except Exception as e:      # noqa: BLE001 -- a late leg must not lose the headline
            import traceback
            res = {"error": f"{type(e).__name__}: {e}"[:600], "traceback": traceback.format_exc()[-1200:]} if rank == 0 else None
            try:
                torch.cuda.empty_cache()
            except Exception:      # noqa: BLE001
                pass
        if rank == 0 and res is not None:
            if isinstance(res, dict):
                res.setdefault("leg_seconds", round(time.perf_counter() - t0, 2))
            where[name] = res
            log(f"[bench] {name}: {json.dumps(res)[:1500]}")
        return res

    def broadcast_table(ctx):
        """ONE broadcast of the read-only table image from the rank that built it (RCCL over xGMI)."""
        if not dist:
            return None
        fence()
        t0 = time.perf_counter()
        nbytes = vdist.broadcast_table_image(ctx, dist, rank, comm_dev, ctx_device=ctx_dev)
        fence()
        dt = time.perf_counter() - t0
        return {"bytes": nbytes, "seconds": dt, "gb_per_s": nbytes / dt / 1e9, "backend": "rccl" if args.backend == "nccl" else "gloo"}

    def haps_to_device(haps):
        """The sample's haplotypes as one device tensor + offsets.  With several ranks they are built ONCE, on rank 0, and handed on with one
        broadcast next to the table image's (every rank building the 3 Gb haplotypes on its own was 11 s and 9 GB of host memory per rank)."""
        if rank == 0:
            lens = np.array([len(h) for h in haps], dtype=np.int64)
        if dist:
            meta = torch.zeros(9, dtype=torch.int64, device=comm_dev)
            if rank == 0:
                meta[0] = len(lens)
                meta[1:1 + len(lens)] = torch.from_numpy(lens)
            dist.broadcast(meta, src=0)
            lens = meta[1:1 + int(meta[0])].cpu().numpy()
        hap_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        d_cat = torch.empty(int(hap_off[-1]), dtype=torch.uint8, device="cuda")
        if rank == 0:
            for h, o in zip(haps, hap_off[:-1]):
                d_cat[int(o):int(o) + len(h)] = torch.from_numpy(h).cuda()
        if dist:
            if args.backend == "nccl":
                dist.broadcast(d_cat, src=0)
            else:
                h_cat = d_cat.cpu()
                dist.broadcast(h_cat, src=0)
                d_cat.copy_(h_cat)
        return d_cat, hap_off

    def generate(d_cat, hap_off, seed, first_read, n_reads, on_ctx=None):
        d_block = torch.empty(n_reads * (READ_LEN + 1), dtype=torch.uint8, device="cuda")
        chunk = 8_000_000
        for first in range(0, n_reads, chunk):
            n = min(chunk, n_reads - first)
            (on_ctx or ctx).synth_reads_device(seed, first_read + first, n, READ_LEN, d_cat, hap_off, d_block[first * (READ_LEN + 1):])
        return d_block

    def timed(step, steps, warmup, min_seconds=1.0, on_ctx=None):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.  A timed
        region shorter than `min_seconds` (the caller's --steps of a 5 ms step) is never the whole measurement: the K-step
        region is then repeated and the reported time is the mean over the repeats -- still K steps' worth, each repeat
        bracketed the same way."""
        c = on_ctx or ctx
        for _ in range(warmup):
            step()
        c.count_kernel_ms()
        kernel_ms, launches, total, repeats = 0.0, 0, 0.0, 0
        while True:
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
                ms, n = c.count_kernel_ms()   # HIP events on the stream the count kernels run on, around the whole count pass of a launch
                kernel_ms += ms
                launches += n
            fence()
            elapsed = time.perf_counter() - t0
            if dist:
                elapsed = vdist.max_over_ranks(elapsed, dist, comm_dev)   # the same number on every rank: same loop exit
            total += elapsed
            repeats += 1
            if total >= min_seconds or repeats >= 1000:
                break
        timed.repeats = repeats
        return total / repeats, kernel_ms / max(launches, 1)

    def per_rank(x):
        """a per-rank figure as (min, max) over the ranks"""
        if not dist:
            return [x, x]
        return [-vdist.max_over_ranks(-float(x), dist, comm_dev), vdist.max_over_ranks(float(x), dist, comm_dev)]

    def verify(keys, d_block, n_check, k=K, d_off=None):
        """An unsaturated prefix of the sample, counter by counter against the oracle (the checker; outside every
        timed region).  A kernel that merely touches every key often enough cannot pass this."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        n_check = min(n_check, d_block.numel() // (READ_LEN + 1))
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, n_check * (READ_LEN + 1), n_check, d_off[: n_check + 1] if d_off is not None else None)
        got, _, _ = ctx.counts_finish()
        t = oracle_lib.Table(keys)
        t.count_block(d_block[: n_check * (READ_LEN + 1)].cpu().numpy(), k)
        want = t.counts()
        ok = bool(np.array_equal(got, want))
        return {"reads": n_check, "oracle_match": ok, "cov_sum": int(got.astype(np.int64).sum()),
                "keys_nonzero": int((got != 0).sum()), "keys_saturated": int((got == 255).sum()),
                "distinct_counter_values": int(np.unique(got).size)}

    def large_kernel():
        """the count pass of graphs that live in HBM as this build and environment run it"""
        name = large_kernel_name()
        if name == "vgk::count27c_kernel" and os.environ.get("VGMI_CT_DEFER", "1") != "0":
            return "vgk::countkc_defer_kernel<27u>"
        return name

    def spread_over_contexts(keys, d_block, n, n_ctx, steps):
        """The count pass's time per launch on FRESH contexts (each its own allocations and table build): the large-table kernels land
        on one of a few times per placement of their memory, so one context's figure is one draw."""
        vals = []
        for _ in range(n_ctx):
            c2 = vgmi.Context(local, buffer_mib=16)
            try:
                if rank == 0 or not dist:
                    c2.table_upload(keys, K)
                else:
                    c2.table_clone_from(ctx)
                d_cov = torch.empty(c2.table_info()["n_keys"], dtype=torch.uint8, device="cuda")
                best = []
                for i in range(steps + 1):
                    c2.counts_reset()
                    c2.reads_submit_device(d_block, n * (READ_LEN + 1), n)
                    c2.counts_finish_device(d_cov, None, None)
                    ms, _ = c2.count_kernel_ms()
                    if i:
                        best.append(ms)
                vals.append(float(np.mean(best)))
                del d_cov
            finally:
                c2.close()
        return vals

    # ================= C2: the 1 Mb graph (table on-chip), BASELINE.json configs[1] -- the headline =================
    g = load_graph() if rank == 0 else None
    if rank == 0:
        ctx.table_upload(g["keys"], g["k"])
    bcast_c2 = broadcast_table(ctx)
    if dist:
        # node CSR + flags: small host-side graph data every rank needs for the gather
        arrs = {k: g[k] for k in ("node_off", "node_key_index", "hom_flag")} if rank == 0 else None
        arrs = vdist.broadcast_arrays(arrs, dist, rank, comm_dev)
        node_off, node_key_index, hom_flag = arrs["node_off"], arrs["node_key_index"], arrs["hom_flag"]
    else:
        node_off, node_key_index, hom_flag = g["node_off"], g["node_key_index"], g["hom_flag"]
    ctx.nodes_upload(node_off, node_key_index)
    ctx.flags_upload(hom_flag)
    info = ctx.table_info()

    haps = cohort_haplotypes() if rank == 0 else None
    d_cat2, hap_off2 = haps_to_device(haps)
    n_reads = args.reads // world if shard else args.reads
    first_read = rank * n_reads if shard else 0
    d_block = generate(d_cat2, hap_off2, 1000 if shard else 1000 + rank, first_read, n_reads)
    n_bytes = n_reads * (READ_LEN + 1)
    d_cov = torch.empty(max(info["n_keys"], 1), dtype=torch.uint8, device="cuda")
    d_cov_node = torch.empty(max(int(node_off[-1]), 1), dtype=torch.uint8, device="cuda")
    d_hist = torch.empty(256, dtype=torch.int64, device="cuda")

    def step():
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, n_bytes, n_reads)
        if shard:
            vdist.allreduce_counts(ctx, dist, comm_dev, ctx_device=ctx_dev)
        ctx.counts_finish_device(d_cov, d_cov_node, d_hist)

    elapsed, kernel_ms = timed(step, args.steps, args.warmup)
    c2_repeats = timed.repeats
    kernel_ms_ranks = per_rank(kernel_ms)
    cov_sum = int(d_cov.to(torch.int64).sum().item())
    hist = d_hist.cpu().numpy()
    # the UNSATURATED rate: the first launch after a reset, short enough that no counter is near the clamp for most of it (the timed
    # steps above run 15 000 x deep: every counter passes 255 within the first few per cent of a launch, and from then on a hit is
    # done at its saturation bit -- no slot fetch, no atomic; that steady state IS the configuration BASELINE names, but not the only one)
    unsat = None
    if args.unsaturated_reads and not shard:
        unsat = []
        for n_u in sorted({min(args.unsaturated_reads, n_reads), min(20_000_000, n_reads)}):
            vals = []
            for _ in range(5):
                ctx.counts_reset()
                ctx.count_kernel_ms()
                ctx.reads_submit_device(d_block, n_u * (READ_LEN + 1), n_u)
                ctx.counts_finish_device(d_cov, d_cov_node, d_hist)
                ms, _ = ctx.count_kernel_ms()
                vals.append(ms)
            unsat.append({"reads": n_u, "kernel_ms": float(np.median(vals)), "kernel_ms_min_max": [min(vals), max(vals)],
                          "keys_saturated_at_the_end": int((d_cov == 255).sum().item()), "keys": int(info["n_keys"])})
    # hits per read on an unsaturated stretch (the counters clamp at 255): measured, for B_probe
    ver_c2 = verify(g["keys"], d_block, args.verify_reads) if rank == 0 and args.verify_reads else None
    del d_block
    torch.cuda.empty_cache()

    if rank == 0:
        total_reads = world * n_reads * args.steps
        value = total_reads / elapsed
        avg_kernel_s = kernel_ms / 1e3
        n_node = int(node_off[-1])
        b_stream = READ_LEN + (info["n_keys"] + n_node) / n_reads   # bases + amortised read-out
        # hits per read for B_probe: measured on the unsaturated verification prefix (the full sample clamps)
        hits_per_read = ver_c2["cov_sum"] / ver_c2["reads"] if ver_c2 and not ver_c2["keys_saturated"] else None
        b_probe = (READ_LEN - K + 1) * 8 + 2 * (hits_per_read if hits_per_read is not None else 0.0)
        ach = b_stream * n_reads / avg_kernel_s / 1e9
        ach_probe = (b_stream + b_probe) * n_reads / avg_kernel_s / 1e9
        traffic, traffic_src = measured_traffic("c2", n_reads, "vgk::count27s_kernel<true, 27u>")
        for u in unsat or []:
            b_u = READ_LEN + (info["n_keys"] + n_node) / u["reads"]
            u["achieved"] = b_u * u["reads"] / (u["kernel_ms"] * 1e-3) / 1e9
            u["frac"] = u["achieved"] / HBM_PEAK_GBS
            u["reads_per_s"] = u["reads"] / (u["kernel_ms"] * 1e-3)
            u["note"] = ("median of 5 first launches after a reset; same B_stream accounting as `frac`; a launch stages its 128 KiB filter once per "
                         "workgroup (~0.3 ms whatever the block's size), and a graph of 5e4 k-mers saturates under ANY sample of this depth: 3e6 reads "
                         "leave no counter at the clamp, 2e7 reads most of them")
        out.update({
            "metric": "150 bp reads/sec genotyped (k=27)",
            "value": value, "unit": "reads/s", "n_gpus": world, "world_size": dist.get_world_size() if dist else 1,
            "steps": args.steps, "warmup": args.warmup,
            "timed_region": {"repeats_of_the_k_step_region": c2_repeats, "seconds_in_all": elapsed * c2_repeats,
                             "note": "a K-step region shorter than 1 s is repeated (each repeat bracketed by barrier + "
                                     "synchronize) and ms_per_step is the mean over the repeats"},
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if shard else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2: 1 Mb ref + 1 k SNP graph (15 haplotypes, 53 734 k-mers), "
                                   f"{n_reads // 2} read pairs 2x150 bp per sample, k=27, one sample per GPU",
                       "reads_per_sample": n_reads, "graph_kmers": info["n_keys"], "table_slots": info["n_slots"],
                       "prefilter_bits": info["filter_bits"], "parallelism": (f"one sample, reads sharded x{world} + all-reduce" if shard else f"sample-per-gpu x{world}")},
            "table_broadcast": bcast_c2,
            "rccl": None if not dist else {"nranks": dist.get_world_size(), "backend": "rccl" if args.backend == "nccl" else "gloo",
                                           "init_process_group_s": t_init, "kernel_ms_min_max_over_ranks": kernel_ms_ranks,
                                           "table_broadcast_gb_per_s": bcast_c2["gb_per_s"] if bcast_c2 else None},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": b_stream * n_reads,
                         "kernel": "vgk::count27s_kernel<true, 27u>", "kernel_ms": avg_kernel_s * 1e3,
                         "unsaturated": unsat,
                         "bytes_per_read": b_stream,
                         "note": "C2's 0.43 MB table is on-chip, so the compulsory HBM traffic is the ASCII read stream only (SURVEY 8d B_stream); "
                                 "the kernel is instruction-issue bound (VALU + LDS; scan 2.7 ms, path-table drain 1.5 ms: DESIGN.md 4.1).  `frac` is the "
                                 "SATURATED steady state of a 15 000 x sample (every counter at the clamp after the first few per cent of a launch: a hit "
                                 "ends at its saturation bit); `unsaturated` lists first launches after a reset (3e6 and 2e7 reads), where hits fetch their slots and "
                                 "bump their counters"},
            "probe_inclusive_rate": {"achieved": ach_probe, "unit": "GB/s", "bytes_per_read": b_stream + b_probe,
                                     "hits_per_read_measured": hits_per_read,
                                     "note": "SURVEY 8d B_stream+B_probe bytes over the same kernel time, for comparison "
                                             "with the c3 block only: the probes are served on-chip here, so this is NOT "
                                             "an HBM fraction"},
            "check": {"cov_sum": cov_sum, "hist_nonzero_bins": int((hist > 0).sum())},
            "verify": ver_c2,
            "library": {"libvgmi_sha256": build.lib_digest(), "source_sha256": build.source_digest()},
        })
        log("[bench] c2: " + json.dumps({k: out[k] for k in ("value", "ms_per_step", "n_gpus")} | {"frac": out["roofline"]["frac"], "kernel_ms": out["roofline"]["kernel_ms"],
                                                                                                   "unsaturated": unsat}))
    del d_cat2

    # the reference's CPU path on this box's cores (a required block of the line: it comes straight behind the headline, ahead of every optional leg)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        run_leg("cpu_baseline", 50, lambda: cpu_baseline(haps, args.cpu_reads, os.cpu_count() or 1))

    # ================= C3 / C4: the chr20-class graph (table in HBM), BASELINE.json configs[2], [3] =================
    state = {}

    def leg_c3():
        from varigraph_amd import synth
        t_g = time.perf_counter()
        keys3, haps3 = synth.snp_graph(args.c3_genome, args.c3_variants) if rank == 0 else (None, None)      # built once; the other ranks
        t_g = time.perf_counter() - t_g                                                                        # receive table image and haplotypes
        if rank == 0:
            ctx.table_upload(keys3, K)
        bcast_c3 = broadcast_table(ctx)
        t_h = time.perf_counter()
        d_cat3, hap_off3 = haps_to_device(haps3)
        t_h = time.perf_counter() - t_h
        info3 = ctx.table_info()
        n3 = args.c3_reads
        d_block3 = generate(d_cat3, hap_off3, 99 + rank, 0, n3)
        del d_cat3
        d_cov3 = torch.empty(info3["n_keys"], dtype=torch.uint8, device="cuda")
        state.update(keys3=keys3, d_block3=d_block3, n3=n3)

        def step3():
            ctx.counts_reset()
            ctx.reads_submit_device(d_block3, n3 * (READ_LEN + 1), n3)
            ctx.counts_finish_device(d_cov3, None, None)

        el3, kms3 = timed(step3, args.c3_steps, 1)
        kms3_ranks = per_rank(kms3)
        cov3 = d_cov3.cpu().numpy()
        del d_cov3
        # the same launch on fresh contexts: median, min, max (this context's own figure is one of the draws)
        draws = [kms3] + (spread_over_contexts(keys3, d_block3, n3, args.c3_contexts - 1, 3) if args.c3_contexts > 1 and not dist else [])
        if rank != 0:
            return None
        # hits per read, measured: the clamped counters of the full sample (30x: nothing near 255) summed
        hits3 = float(cov3.astype(np.int64).sum()) / n3
        b_read3 = READ_LEN + (READ_LEN - K + 1) * 8 + 2 * hits3 + info3["n_keys"] / n3
        med3 = float(np.median(draws))
        ach3 = b_read3 * n3 / (med3 * 1e-3) / 1e9
        tr3, tr3_src = measured_traffic("c3", n3, large_kernel())
        return {"workload": f"C3: chr20-class synthetic SNP graph ({args.c3_genome // 1_000_000} Mb, {args.c3_variants} SNPs, {info3['n_keys']} k-mers), "
                            f"{n3 // 2} read pairs 2x150 bp per sample, one sample per GPU",
                "value": world * n3 * args.c3_steps / el3, "unit": "reads/s", "steps": args.c3_steps,
                "ms_per_step": el3 / args.c3_steps * 1e3, "graph_kmers": info3["n_keys"], "table_slots": info3["n_slots"],
                "graph_build_s": t_g, "table_broadcast": bcast_c3, "haplotypes_to_the_ranks_s": t_h,
                "context_table": ctx.ctable_info() if ctx.ctable_info()["n_buckets"] else None,
                "hits_per_read": hits3, "keys_saturated": int((cov3 == 255).sum()),
                "roofline": {"bound": "hbm", "achieved": ach3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach3 / HBM_PEAK_GBS, "traffic": tr3, "traffic_source": tr3_src,
                             "algorithmic_bytes_per_launch": b_read3 * n3, "bytes_per_read": b_read3,
                             "kernel": large_kernel(),
                             "kernel_ms": med3, "kernel_ms_min_max": [min(draws), max(draws)], "kernel_ms_by_context": draws,
                             "kernel_ms_min_max_over_ranks": kms3_ranks,
                             "frac_min_max": [b_read3 * n3 / (max(draws) * 1e-3) / 1e9 / HBM_PEAK_GBS, b_read3 * n3 / (min(draws) * 1e-3) / 1e9 / HBM_PEAK_GBS],
                             "note": "SURVEY 8d: B_read = 150 (bases) + 124 x 8 (one key compare per k-mer) + 2 x hits + amortised read-out; hits measured "
                                     "from this run's counters.  kernel_ms = HIP events around the WHOLE count pass of a launch (the count kernel, since round 6 "
                                     "the two kernels that add its runs of hits up by counter region, the ragged tail), median over the timed context and "
                                     f"{max(0, len(draws) - 1)} fresh ones"},
                "verify": verify(keys3, d_block3, args.verify_reads) if args.verify_reads else None}

    def leg_other_k():
        # the same reads over the graph built for other k (the context table with flanks of k - 16 bases; even k behind the pass that takes
        # back what the reference's run counter suppresses; k = 28 since round 6): rates next to the k = 27 one above, each oracle-checked
        from varigraph_amd import synth
        d_block3, n3 = state["d_block3"], state["n3"]
        other = {}
        for kk in (25, 21, 22, 28):
            keys_k, _ = synth.snp_graph(args.c3_genome, args.c3_variants, k=kk)
            ctx.table_upload(keys_k, kk)
            d_off_k = (torch.arange(n3 + 1, dtype=torch.int64, device="cuda") * (READ_LEN + 1)) if kk % 2 == 0 else None
            d_cov_k = torch.empty(len(keys_k), dtype=torch.uint8, device="cuda")

            def step_k():
                ctx.counts_reset()
                ctx.reads_submit_device(d_block3, n3 * (READ_LEN + 1), n3, d_off_k)
                ctx.counts_finish_device(d_cov_k, None, None)

            el_k, kms_k = timed(step_k, 2, 1, min_seconds=0.2)
            cinfo_k = ctx.ctable_info()
            other[str(kk)] = {"reads_per_s": n3 * 2 / el_k, "kernel_ms": kms_k, "graph_kmers": int(len(keys_k)),
                              "context_table_buckets": cinfo_k["n_buckets"],
                              "verify": verify(keys_k, d_block3, min(args.verify_reads, 200_000), kk, d_off_k) if args.verify_reads else None}
            del d_cov_k
        return other

    if not args.no_c3 and not shard:
        run_leg("c3", 25, leg_c3)
        if rank == 0 and world == 1 and not args.no_other_k and isinstance(out.get("c3"), dict) and "roofline" in out["c3"]:
            run_leg("other_k", 30, leg_other_k, where=out["c3"])
        state.clear()
        torch.cuda.empty_cache()

    # ================= C5: the whole-genome-class graph, BASELINE.json configs[4] (single-GPU slice per rank) =================
    def leg_c5():
        from varigraph_amd import synth
        torch.cuda.empty_cache()
        t_g = time.perf_counter()
        keys5, haps5 = synth.snp_graph(args.c5_genome, args.c5_variants) if rank == 0 else (None, None)
        t_g = time.perf_counter() - t_g
        t_u = time.perf_counter()
        if rank == 0:
            ctx.table_upload(keys5, K)
        torch.cuda.synchronize()
        t_u = time.perf_counter() - t_u
        bcast_c5 = broadcast_table(ctx)
        t_h = time.perf_counter()
        d_cat5, hap_off5 = haps_to_device(haps5)
        t_h = time.perf_counter() - t_h
        del haps5
        info5, xinfo5, cinfo5 = ctx.table_info(), ctx.xtable_info(), ctx.ctable_info()
        n5 = args.c5_reads
        d_block5 = generate(d_cat5, hap_off5, 4711 + rank, 0, n5)
        del d_cat5
        d_cov5 = torch.empty(info5["n_keys"], dtype=torch.uint8, device="cuda")

        def step5():
            ctx.counts_reset()
            ctx.reads_submit_device(d_block5, n5 * (READ_LEN + 1), n5)
            ctx.counts_finish_device(d_cov5, None, None)

        el5, kms5 = timed(step5, args.c5_steps, 1)
        kms5_ranks = per_rank(kms5)
        cov5 = d_cov5.cpu().numpy() if rank == 0 else None
        del d_cov5
        free_b, total_b = ctx.device_memory()
        draws = [kms5] + (spread_over_contexts(keys5, d_block5, n5, args.c5_contexts - 1, 2) if args.c5_contexts > 1 and not dist else [])
        if rank != 0:
            return None
        hits5 = float(cov5.astype(np.int64).sum()) / n5
        b_read5 = READ_LEN + (READ_LEN - K + 1) * 8 + 2 * hits5 + info5["n_keys"] / n5
        med5 = float(np.median(draws))
        ach5 = b_read5 * n5 / (med5 * 1e-3) / 1e9
        ver5 = None
        if args.verify_reads:
            # the oracle's emitted keys of a prefix (vgo_sketch: the reference's state machine), looked up by binary
            # search in the sorted key list -- a CPU hash table over 2.7e8 keys is not needed for the check
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            m = min(100_000, args.verify_reads, n5)
            ctx.counts_reset()
            ctx.reads_submit_device(d_block5, m * (READ_LEN + 1), m)
            got5, _, _ = ctx.counts_finish()
            rows = d_block5[: m * (READ_LEN + 1)].cpu().numpy().reshape(m, READ_LEN + 1)
            emitted = np.concatenate([oracle_lib.sketch(rows[i, :READ_LEN].tobytes(), K) for i in range(m)])
            pos = np.searchsorted(keys5, emitted)      # np.unique output: sorted
            pos[pos == keys5.size] = 0
            idx, cnt = np.unique(pos[keys5[pos] == emitted], return_counts=True)
            want5 = np.zeros(keys5.size, dtype=np.uint8)
            want5[idx] = np.minimum(cnt, 255).astype(np.uint8)
            ver5 = {"reads": m, "oracle_match": bool(np.array_equal(got5, want5)), "cov_sum": int(got5.astype(np.int64).sum()),
                    "keys_nonzero": int((got5 != 0).sum()), "method": "oracle sketch per read + binary search in the sorted key list"}
        tr5, tr5_src = measured_traffic("c5", n5, large_kernel_name())
        return {"workload": f"C5 single-GPU slice: whole-genome-class synthetic SNP graph ({args.c5_genome / 1e9:g} Gb, {args.c5_variants} SNPs, {info5['n_keys']} k-mers, "
                            f"graph index HBM-resident), {n5 // 2} read pairs 2x150 bp per sample, one sample per GPU",
                "value": world * n5 * args.c5_steps / el5, "unit": "reads/s", "steps": args.c5_steps,
                "ms_per_step": el5 / args.c5_steps * 1e3, "graph_kmers": info5["n_keys"], "table_slots": info5["n_slots"],
                "large_table_gb": (xinfo5["n_lines"] * 128 + cinfo5["n_buckets"] * 64) / 1e9,
                "context_table": cinfo5 if cinfo5["n_buckets"] else None,
                "xtable_overflow_pairs": xinfo5["overflow_pairs"],
                "device_memory_in_use_gb": (total_b - free_b) / 1e9,
                "graph_build_s": t_g, "table_upload_s": t_u, "table_broadcast": bcast_c5, "haplotypes_to_the_ranks_s": t_h, "hits_per_read": hits5,
                "keys_saturated": int((cov5 == 255).sum()),
                "roofline": {"bound": "hbm", "achieved": ach5, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach5 / HBM_PEAK_GBS,
                             "traffic": tr5, "traffic_source": tr5_src, "algorithmic_bytes_per_launch": b_read5 * n5, "bytes_per_read": b_read5,
                             "kernel": large_kernel_name(), "kernel_ms": med5, "kernel_ms_min_max": [min(draws), max(draws)], "kernel_ms_by_context": draws,
                             "kernel_ms_min_max_over_ranks": kms5_ranks,
                             "note": "same accounting as the c3 block (SURVEY 8d); a table of this many counters keeps its counter updates in the row loop "
                                     "(vgmi_ctdefer.hip serves up to 6.7e7 counters)"},
                "verify": ver5}

    if not args.no_c5 and not shard:
        run_leg("c5", 75, leg_c5)
        torch.cuda.empty_cache()

    # ================= construct side: K3 counting-Bloom update and K4 query (15 B per reference k-mer) =================
    def leg_bloom():
        G = 60_000_000                                   # the chr20-class reference of config 3 / 4
        m_b, nh_b = vgmi.bloom_params(G - K + 1, 0.01)
        gen = torch.Generator(device="cuda").manual_seed(7)
        seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (G,), generator=gen, device="cuda")]
        seeds_b = np.arange(1, nh_b + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        best_add = None
        for _ in range(4):
            ctx.bloom_create(m_b, nh_b, seeds_b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.bloom_add_seq_device(seq, G, K)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best_add = dt if best_add is None or dt < best_add else best_add
        qk = (np.random.default_rng(3).integers(0, 1 << 54, size=10_000_000, dtype=np.uint64) << np.uint64(8)) | np.uint64(K)
        t0 = time.perf_counter()
        mn, nz = ctx.bloom_query(qk)                     # host keys in, host answers out (PCIe-inclusive, as construct calls it)
        t_q = time.perf_counter() - t0
        n_km = G - K + 1
        binned = os.environ.get("VGMI_BLOOM_BINNED", "1") != "0"
        tb, tb_src = measured_traffic("bloom", n_km, None) if binned else (None, None)
        # the binned form's own traffic, by construction: the k-mer keys written and read (8 B a position), the positions written and
        # read at both levels of the partition (4 B each), the filter read and written once
        est = 16.0 * G + 16.0 * nh_b * n_km + 2.0 * m_b
        del seq
        torch.cuda.empty_cache()
        return {"workload": f"K3: every k-mer of a {G // 1_000_000} Mb random reference (resident in HBM) into BloomFilter(n = G - k + 1, p = 0.01): "
                            f"{m_b / 1e6:.0f} MB of saturating byte counters, {nh_b} MurmurHash3 positions per k-mer; K4: 1e7 random keys queried",
                "add_kmers_per_s": n_km / best_add, "add_seconds": best_add,
                "filter_updates_per_s": nh_b * n_km / best_add,
                "form": "binned by 128 KiB filter chunk, counted in LDS (vgmi_bloom_bin.hip)" if binned else "direct compare-and-swaps (VGMI_BLOOM_BINNED=0)",
                "roofline": {"bound": "hbm", "achieved": 15.0 * n_km / best_add / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": 15.0 * n_km / best_add / 1e9 / HBM_PEAK_GBS, "traffic": tb, "traffic_source": tb_src,
                             "traffic_by_construction": est if binned else None,
                             "bytes_per_kmer": 15.0,
                             "kernel": "vgk::bb_scatter1_kernel + bb_scatter2_kernel + bb_accumulate_kernel (+ rows_kernel<1, false> for the k-mers)" if binned
                                       else "vgk::rows_kernel<2, false>",
                             "note": ("SURVEY 8d accounting: 1 base + 7 byte read-modify-writes per k-mer, over the whole call.  The direct form is bound by "
                                      "the device's atomic rate (2.3e10 compare-and-swaps a second of 2.7e10); the binned form moves several times the accounted "
                                      "bytes as streams instead: DESIGN.md 4.3b") if binned else
                                     ("SURVEY 8d accounting: 1 base + 7 byte read-modify-writes per k-mer.  The kernel is bound by random "
                                      "32-bit compare-and-swaps on a filter far larger than the caches (one 64-byte sector per "
                                      "byte counter), not by bytes: see DESIGN.md section 6")},
                "query_keys_per_s_pcie_inclusive": qk.size / t_q, "query_hits": int(nz.sum())}

    if rank == 0 and world == 1 and not args.no_bloom:
        run_leg("bloom", 10, leg_bloom)

    if rank == 0 and world == 1 and not args.no_sample_level:
        run_leg("sample_level", 45, lambda: sample_level(ctx, haps, out.get("cpu_baseline") if isinstance(out.get("cpu_baseline"), dict) else None))
    ctx.close()
    if rank == 0:
        for w in ("c3", "c5"):      # (the configurations whose table lives in HBM, inside the object the driver's record keeps: their own blocks carry the rest)
            b = out.get(w)
            if isinstance(b, dict) and "roofline" in b:
                out["roofline"][f"{w}_frac_kernel"] = b["roofline"]["frac"]
                out["roofline"][f"{w}_kernel_ms"] = b["roofline"]["kernel_ms"]
                out["roofline"][f"{w}_kernel_ms_min_max"] = b["roofline"]["kernel_ms_min_max"]
        if world == 1 and not args.no_c4:
            torch.cuda.empty_cache()
            run_leg("c4", 20 + 3 * args.c4_pairs / 1e6, lambda: c4_cli(args.c4_samples, 60_000_000, 500_000, args.c4_pairs, args.c4_threads))
        out["wall_s"] = round(time.perf_counter() - t_start, 1)
        emit()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
