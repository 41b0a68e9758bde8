#!/usr/bin/env python3
"""CLI-level wall time (SURVEY 8d metric level iii, for information): BASELINE.json config 1
(1 Mb graph, 100 k pairs 2x150 bp) through `genotype` of (a) the all-CPU reference, (b) the
integration build (reference code, read counting on the MI355X) and (c) the native CLI varigraph-mi (no
reference code); the VCFs must be identical."""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    from varigraph_amd import synth, vgmi
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "c1", "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    variants, gts = synth.make_cohort(ref, meta["n_var"], n_samples=meta["n_samples"], ploidy=meta["ploidy"],
                                      seed=meta["cohort_seed"])
    haps = synth.sample_haplotypes(ref, variants, gts, 0, meta["ploidy"])
    work = tempfile.mkdtemp(prefix="vg_cli_")
    out = {"n_pairs": n_pairs}
    try:
        block = vgmi.synth_reads_host(1000, 0, 2 * n_pairs, 150, haps)
        fq = synth.write_fastq_pair(os.path.join(work, "s"), block, 2 * n_pairs, 150, gz=True)
        graph = os.path.join(work, "graph.bin")
        with open(graph, "wb") as f:
            f.write(gzip.open(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"), "rb").read())
        vcfs = {}
        exes = (("reference_cpu", os.path.join(ROOT, "oracle", "_ref", "varigraph_det"), []),
                ("integration_hip", os.path.join(ROOT, "oracle", "_ref", "varigraph_hip"), ["--gpu", "0"]),
                ("native_cli", os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi"), ["--gpu", "0"]))
        env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
        for name, exe, extra in exes:
            if not os.path.exists(exe):
                continue
            d = os.path.join(work, name)
            os.makedirs(d)
            open(os.path.join(d, "samples.cfg"), "w").write("sample0 " + " ".join(fq) + "\n")
            t0 = time.perf_counter()
            r = subprocess.run([exe, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", "10"] + extra, cwd=d,
                               capture_output=True, text=True, env=env)
            out[name + "_wall_s"] = time.perf_counter() - t0
            if r.returncode != 0:
                out[name + "_error"] = r.stderr[-300:]
                continue
            vcfs[name] = gzip.open(os.path.join(d, "sample0.varigraph.vcf.gz"), "rb").read()
        out["vcf_identical"] = len(vcfs) >= 2 and all(v == vcfs["reference_cpu"] for v in vcfs.values())
        out["binaries"] = sorted(vcfs)
        out["vcf_records"] = vcfs.get("reference_cpu", b"").count(b"\n")
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
