#!/usr/bin/env python3
"""Which block-gzip member does the device inflate refuse, and why (status code)?  Diagnostic for vgmi_inflate.hip."""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    import tempfile, shutil
    from varigraph_amd import host, synth, vgmi
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "c1", "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    variants, gts = synth.make_cohort(ref, meta["n_var"], n_samples=meta["n_samples"], ploidy=meta["ploidy"], seed=meta["cohort_seed"])
    haps = synth.sample_haplotypes(ref, variants, gts, 0, meta["ploidy"])
    work = tempfile.mkdtemp(prefix="vg_dbg_")
    try:
        block = vgmi.synth_reads_host(1000, 0, n_reads, 150, haps)
        plain = synth.write_fastq_pair_fast(os.path.join(work, "s"), block, n_reads, 150)
        bgz = synth.bgzf_compress_file(plain[0], plain[0] + ".bgz.gz", level=level)
        comp = open(bgz, "rb").read()
        g = host.Graph(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"))
        ctx = vgmi.Context(0, buffer_mib=64)
        g.upload(ctx)
        ctx.counts_reset()
        r = ctx.fastq_bgzf(comp)
        r.pop("tail")
        offs, pos = [], 0
        while pos < len(comp):
            offs.append(pos)
            pos += (comp[pos + 16] | comp[pos + 17] << 8) + 1
        r["members"] = len(offs)
        if r["inflate_failed"]:
            bad = offs.index(r["good_compressed_bytes"]) if r["good_compressed_bytes"] in offs else -1
            r["first_bad_member"] = bad
            if bad >= 0:
                m = comp[offs[bad]:offs[bad + 1]] if bad + 1 < len(offs) else comp[offs[bad]:]
                d = zlib.decompressobj(-15)
                text = d.decompress(m[18:-8])
                r["bad_member_text_len"] = len(text)
                r["bad_member_comp_len"] = len(m)
                # block structure of the member: walk with zlib's Z_BLOCK
                d2 = zlib.decompressobj(-15)
                r["bad_member_head"] = text[:80].decode(errors="replace")
        print(json.dumps(r))
        ctx.counts_finish()
        ctx.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)
if __name__ == "__main__":
    main()
