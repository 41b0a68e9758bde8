#!/usr/bin/env python3
"""BASELINE.json configs[4] through the product CLI at (scaled) whole-genome class: `varigraph-mi construct` + `varigraph-mi genotype
--sample-ploidy 4 --use-depth` on a synthetic cohort of SURVEY 8d's shape -- several contigs, SNPs + indels + long insertions, a few
tetraploid VCF samples -- one sample's read pairs drawn by the device generator and STREAMED into the CLI through two named pipes
(the CLI reads plain FASTQ sequentially: no file of tens of gigabytes is written).

The all-CPU reference cannot be run at this size in test time, so what is checked is
  * the counters of a read prefix, through the graph the CLI wrote and the C ABI, against the oracle's emitted keys (as
    tests/test_gpu_large.py::test_c5_wgs_class_single_gpu_slice does for the synthetic key set),
  * the called genotypes against the generator's truth (dosage of the ALT allele per site), and
  * what the run cost: wall per stage from the CLI's own log (VGH_TIMING=1), peak RSS of construct and genotype.
The oracle (tests/oracle_lib.py) is the checker here, never the thing measured.

  python tools/wgs_cli_e2e.py --genome 3000000000 --contigs 24 --variants 5000000 --pairs 100000000 > profiles/r4_e2e_wgs_tetraploid.json
"""
import argparse
import gzip
import json
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = 150


def mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) / 1e6
                break
        try:
            mx = open("/sys/fs/cgroup/memory.max").read().strip()
            if mx != "max":
                avail = min(avail, int(mx) / 1e9)
        except OSError:
            pass
        return avail
    except Exception:
        return None


def write_fasta_fast(f, name, ref, width=60):
    f.write(b">" + name.encode() + b"\n")
    n = len(ref) // width * width
    if n:
        m = np.empty((n // width, width + 1), dtype=np.uint8)
        m[:, :width] = ref[:n].reshape(-1, width)
        m[:, width] = 10
        f.write(m.tobytes())
    if n < len(ref):
        f.write(ref[n:].tobytes() + b"\n")


def child_rss_gb():
    return resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6      # KiB -> GB (maximum over the children waited for so far)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=300_000_000)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--variants", type=int, default=500_000)
    ap.add_argument("--pairs", type=int, default=10_000_000)
    ap.add_argument("--vcf-samples", type=int, default=3)
    ap.add_argument("--ploidy", type=int, default=4)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--indel", type=float, default=0.05)
    ap.add_argument("--sv", type=float, default=0.001)
    ap.add_argument("--prefix-reads", type=int, default=100_000)
    ap.add_argument("--files", action="store_true", help="write the FASTQ files instead of streaming them through named pipes")
    ap.add_argument("--bgzf", action="store_true", help="with --files: block-gzip files (level 1, 64 KiB members, compressed by a pool of threads)")
    ap.add_argument("--samples", type=int, default=1, help="sample names in the -s list; they all name the same two files (round 5: several whole-genome "
                                                              "samples back to back through one process)")
    ap.add_argument("--keep", default="")
    args = ap.parse_args()
    import torch
    from varigraph_amd import host, synth, vgmi
    cli = os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi")
    out = {"genome": args.genome, "contigs": args.contigs, "variants": args.variants, "pairs": args.pairs, "ploidy": args.ploidy,
           "vcf_samples": args.vcf_samples, "threads": args.threads, "mem_available_gb_at_start": mem_available_gb()}
    need = 4e-9 * args.genome * (1 + args.ploidy) + 10e-6 * args.variants      # haplotypes (twice while they are joined) + graph structures, GB: measured 3.6 GB at 300 Mb / 5e5 variants
    if out["mem_available_gb_at_start"] is not None and out["mem_available_gb_at_start"] < need:
        out["error"] = f"about {need:.0f} GB of host memory wanted, {out['mem_available_gb_at_start']:.0f} available: not run"
        print(json.dumps(out))
        return
    work = args.keep or tempfile.mkdtemp(prefix="vg_wgs_")
    os.makedirs(work, exist_ok=True)
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022", VGH_TIMING="1")
    try:
        # ---- the cohort, contig by contig
        t0 = time.perf_counter()
        fa, vcf = os.path.join(work, "ref.fa"), os.path.join(work, "in.vcf")
        clen = args.genome // args.contigs
        nvar = args.variants // args.contigs
        hap_parts = [[] for _ in range(args.ploidy)]
        truth = {}                                            # (chrom, pos1) -> ALT dosage of sample 0
        with open(fa, "wb") as ff, open(vcf, "w") as fv:
            fv.write("##fileformat=VCFv4.2\n")
            for c in range(args.contigs):
                fv.write(f"##contig=<ID=chr{c + 1},length={clen}>\n")
            fv.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
            fv.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(args.vcf_samples)) + "\n")
            for c in range(args.contigs):
                ref = synth.make_reference(clen, seed=synth.REF_SEED + 17 * c)
                var, gts = synth.make_cohort(ref, nvar, n_samples=args.vcf_samples, ploidy=args.ploidy, seed=11 + c, indel_frac=args.indel,
                                             sv_frac=args.sv)
                write_fasta_fast(ff, f"chr{c + 1}", ref)
                pl = args.ploidy
                lines = []
                for vi, (p, ra, aa) in enumerate(var):
                    g = gts[vi]
                    cols = "\t".join("|".join(map(str, g[s * pl:(s + 1) * pl].tolist())) for s in range(args.vcf_samples))
                    lines.append(f"chr{c + 1}\t{p + 1}\tv{c}_{vi}\t{ra.decode()}\t{aa.decode()}\t.\tPASS\t.\tGT\t{cols}\n")
                    truth[(c + 1, p + 1)] = int(g[:pl].sum())
                fv.write("".join(lines))
                for h in range(pl):
                    hap_parts[h].append(synth.haplotype(ref, var, gts, h))
                del ref, var, gts, lines
        haps = [np.concatenate(x) for x in hap_parts]
        del hap_parts
        out["cohort_s"] = time.perf_counter() - t0
        # ---- construct
        graph = os.path.join(work, "graph.bin")
        t0 = time.perf_counter()
        r = subprocess.run([cli, "construct", "-r", fa, "-v", vcf, "--save-graph", graph, "-t", "32", "--gpu", "0", "--vcf-ploidy", str(args.ploidy)],
                           cwd=work, capture_output=True, text=True, env=env)
        out["construct_s"] = time.perf_counter() - t0
        out["construct_peak_rss_gb"] = child_rss_gb()
        if r.returncode != 0:
            out["error"] = "construct: " + r.stderr[-600:]
            print(json.dumps(out))
            return
        out["graph_bytes"] = os.path.getsize(graph)
        out["construct_log"] = [ln for ln in r.stderr.strip().split("\n") if "varigraph-mi]" in ln][-6:]
        # ---- the reads: device generator -> two named pipes (or files) -> `genotype`
        ctx = vgmi.Context(0, buffer_mib=16)
        off = np.concatenate([[0], np.cumsum([h.size for h in haps])]).astype(np.uint64)
        d_cat = torch.empty(int(off[-1]), dtype=torch.uint8, device="cuda")
        o = 0
        for h in haps:
            d_cat[o:o + h.size] = torch.from_numpy(h).cuda()
            o += h.size
        del haps
        fq = [os.path.join(work, "s_1.fq"), os.path.join(work, "s_2.fq")]
        n_reads = 2 * args.pairs
        chunk = 2_000_000
        d_block = torch.empty(chunk * (L + 1), dtype=torch.uint8, device="cuda")
        prefix_block = []

        def records(rec, first, mate):
            rows = rec[mate::2, :L]
            k = rows.shape[0]
            m = np.empty((k, 14 + L + 3 + L + 1), dtype=np.uint8)
            m[:, 0], m[:, 1] = ord("@"), ord("r")
            idx = np.arange(first // 2, first // 2 + k, dtype=np.int64)
            for d in range(9):
                m[:, 2 + d] = (idx // 10 ** (8 - d)) % 10 + ord("0")
            m[:, 11], m[:, 12], m[:, 13] = ord("/"), ord("1") + mate, 10
            m[:, 14:14 + L] = rows
            m[:, 14 + L], m[:, 15 + L], m[:, 16 + L] = 10, ord("+"), 10
            m[:, 17 + L:17 + 2 * L] = ord("I")
            m[:, 17 + 2 * L] = 10
            return m.tobytes()

        import struct
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max(2, args.threads)) if args.bgzf else None

        def bgzf_member(d):
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            cd = c.compress(d) + c.flush()
            return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(cd) + 25) + cd + struct.pack("<II", zlib.crc32(d), len(d)))

        def bgzf_bytes(b):
            mv = memoryview(b)
            blocks = [mv[i:i + 0xff00] for i in range(0, len(mv), 0xff00)]
            return b"".join(pool.map(bgzf_member, blocks, chunksize=64))

        def produce(files):
            for first in range(0, n_reads, chunk):
                n = min(chunk, n_reads - first)
                ctx.synth_reads_device(1000, first, n, L, d_cat, off, d_block)
                torch.cuda.synchronize()
                rec = d_block[: n * (L + 1)].cpu().numpy().reshape(n, L + 1)
                if first == 0:
                    prefix_block.append(rec[: min(n, args.prefix_reads)].copy().reshape(-1))
                a, b = records(rec, first, 0), records(rec, first, 1)
                if args.bgzf:
                    a, b = bgzf_bytes(a), bgzf_bytes(b)
                ts = [threading.Thread(target=files[0].write, args=(a,)), threading.Thread(target=files[1].write, args=(b,))]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
            if args.bgzf:
                eof = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0"
                for f in files:
                    f.write(eof)
            for f in files:
                f.close()

        if args.bgzf:
            fq = [p + ".gz" for p in fq]
        d = os.path.join(work, "run")
        os.makedirs(d)
        open(os.path.join(d, "samples.cfg"), "w").write("".join(f"sample{i} " + " ".join(fq) + "\n" for i in range(args.samples)))
        cmd = [cli, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", str(args.threads), "--gpu", "0"]
        if args.ploidy != 2:
            cmd += ["--sample-ploidy", str(args.ploidy), "--use-depth"]
        t0 = time.perf_counter()
        if args.files:
            produce([open(p, "wb") for p in fq])
            out["fastq_files_s"] = time.perf_counter() - t0
            out["fastq_file_bytes"] = [os.path.getsize(p) for p in fq]
            out["samples"] = args.samples
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, env=env)
        else:
            for p in fq:
                os.mkfifo(p)
            pr = subprocess.Popen(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
            files = [None, None]

            def opener(i):
                files[i] = open(fq[i], "wb")
            ts = [threading.Thread(target=opener, args=(i,)) for i in (0, 1)]       # a pipe's open waits for its reader
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            prod_err = []

            def run_producer():
                try:
                    produce(files)
                except Exception as e:      # the reader went away
                    prod_err.append(repr(e))
            tp = threading.Thread(target=run_producer)
            tp.start()
            so, se = pr.communicate()
            tp.join()
            r = subprocess.CompletedProcess(cmd, pr.returncode, so, se)
            if prod_err:
                out["producer_error"] = prod_err[0]
        out["genotype_wall_s"] = time.perf_counter() - t0
        out["genotype_peak_rss_gb"] = child_rss_gb()
        out["reads_streamed_through_pipes"] = not args.files
        if r.returncode != 0:
            out["error"] = "genotype: " + r.stderr[-800:]
            print(json.dumps(out))
            return
        log = [ln for ln in r.stderr.strip().split("\n") if "varigraph-mi]" in ln or "graph_index]" in ln]
        out["genotype_log"] = [ln for ln in log if "HMM part" not in ln][-(60 + 12 * args.samples):]
        import re
        cnt = [(float(a), float(b)) for a, b in re.findall(r"counting ([\d.]+) s \(kernel ([\d.]+) s", r.stderr)]
        if cnt:
            out["counting_wall_s_per_sample"] = [c[0] for c in cnt]
            out["count_kernel_s_per_sample"] = [c[1] for c in cnt]
            out["counting_reads_per_s"] = [n_reads / c[0] for c in cnt]
        out["genotyping_wall_s_per_sample"] = [float(x) for x in re.findall(r"genotyping ([\d.]+) s", r.stderr)]
        # ---- called genotypes against the generator's truth
        called = same = het_ok = 0
        with gzip.open(os.path.join(d, "sample0.varigraph.vcf.gz"), "rt") as f:
            for ln in f:
                if ln[0] == "#":
                    continue
                t = ln.split("\t", 10)
                gt = t[9].split(":", 1)[0].replace("|", "/").split("/")
                if "." in gt:
                    continue
                key = (int(t[0][3:]), int(t[1]))
                if key not in truth:
                    continue
                called += 1
                dose = sum(1 for a in gt if a != "0")
                same += dose == truth[key]
                het_ok += (dose > 0) == (truth[key] > 0)
        out["sites"] = len(truth)
        out["sites_called"] = called
        out["dosage_concordance"] = same / max(called, 1)
        out["carrier_concordance"] = het_ok / max(called, 1)
        # ---- a read prefix through the graph the CLI wrote, against the oracle's emitted keys
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        g = host.Graph(graph)
        keys = g.arrays()["keys"].copy()
        order = np.argsort(keys)
        sk = keys[order]
        ctx.table_upload(keys, 27)
        blk = prefix_block[0]
        m = blk.size // (L + 1)
        ctx.counts_reset()
        ctx.reads_submit(blk, m)
        got, _, _ = ctx.counts_finish()
        rows = blk.reshape(m, L + 1)
        emitted = np.concatenate([oracle_lib.sketch(rows[i, :L].tobytes(), 27) for i in range(m)])
        pos = np.searchsorted(sk, emitted)
        pos[pos == sk.size] = 0
        hit = sk[pos] == emitted
        want = np.zeros(keys.size, dtype=np.int64)
        np.add.at(want, order[pos[hit]], 1)
        out["prefix_reads"] = m
        out["prefix_counters_equal_oracle"] = bool(np.array_equal(got, np.minimum(want, 255).astype(np.uint8)))
        out["prefix_hits"] = int(hit.sum())
        out["graph_kmers"] = int(keys.size)
        out["context_table"] = ctx.ctable_info()
        ctx.close()
    finally:
        if not args.keep:
            shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
