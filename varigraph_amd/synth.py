"""Seeded synthetic cohorts for tests and bench.

The reference ships no data, so every workload is generated: an iid
uniform reference (native mixer, vg_synth.h), uniformly placed variants, phased VCF samples
whose first sample is heterozygous everywhere, and reads drawn from that sample's haplotypes
by the native generator (vgmi_synth_reads_host / _device).
"""
import gzip
import os

import numpy as np

from . import vgmi

REF_SEED = 20241022
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def make_reference(length, seed=REF_SEED):
    return vgmi.synth_reference(seed, length)


def make_cohort(ref, n_variants, n_samples=7, ploidy=2, seed=1, indel_frac=0.0, sv_frac=0.0, margin=100):
    """Returns (variants, gts): variants = list of (pos0, ref_allele bytes, alt_allele bytes) sorted,
    non-overlapping; gts = uint8 [n_variants, n_samples*ploidy] allele index (0/1) per haplotype.
    Sample 0 carries 0 on its first ploidy/2 haplotypes and 1 on the rest (all-het)."""
    rng = np.random.default_rng(seed)
    L = len(ref)
    # spaced positions so that variants (incl. deletions up to 20 bp) never overlap
    min_gap = 32
    slots = (L - 2 * margin) // min_gap
    assert n_variants <= slots, "too many variants for this reference"
    chosen = np.sort(rng.choice(slots, size=n_variants, replace=False))
    pos = margin + chosen * min_gap + rng.integers(0, min_gap - 24, size=n_variants)
    kind = rng.random(n_variants)
    variants = []
    for p, u in zip(pos.tolist(), kind.tolist()):
        rc = int(_CODE[ref[p]])
        if u < sv_frac:  # long insertion 60..300 bp
            n = int(rng.integers(60, 301))
            ins = _ACGT[rng.integers(0, 4, size=n)].tobytes()
            variants.append((p, bytes([ref[p]]), bytes([ref[p]]) + ins))
        elif u < sv_frac + indel_frac:
            n = int(rng.integers(1, 21))
            if rng.random() < 0.5:  # insertion
                ins = _ACGT[rng.integers(0, 4, size=n)].tobytes()
                variants.append((p, bytes([ref[p]]), bytes([ref[p]]) + ins))
            else:  # deletion
                variants.append((p, ref[p:p + 1 + n].tobytes(), bytes([ref[p]])))
        else:  # SNP
            alt = int(_ACGT[(rc + int(rng.integers(1, 4))) % 4])
            variants.append((p, bytes([ref[p]]), bytes([alt])))
    gts = rng.integers(0, 2, size=(n_variants, n_samples * ploidy)).astype(np.uint8)
    gts[:, : ploidy // 2 if ploidy > 1 else 0] = 0
    gts[:, max(ploidy // 2, 0):ploidy] = 1
    if ploidy == 1:
        gts[:, 0] = 1
    return variants, gts


def haplotype(ref, variants, gts, hap_col):
    """Apply the ALT alleles carried by haplotype column `hap_col`."""
    out = []
    prev = 0
    for (p, ra, aa), g in zip(variants, gts[:, hap_col]):
        if g:
            out.append(ref[prev:p])
            out.append(np.frombuffer(aa, dtype=np.uint8))
            prev = p + len(ra)
    out.append(ref[prev:])
    return np.ascontiguousarray(np.concatenate(out))


def write_fasta(path, chrom, ref, width=60):
    with open(path, "wb") as f:
        f.write(b">" + chrom.encode() + b"\n")
        b = ref.tobytes()
        for i in range(0, len(b), width):
            f.write(b[i:i + width] + b"\n")


def write_vcf(path, chrom, ref_len, variants, gts, n_samples, ploidy):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wt") as f:
        f.write("##fileformat=VCFv4.2\n")
        f.write(f"##contig=<ID={chrom},length={ref_len}>\n")
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
                "\t".join(f"S{i}" for i in range(n_samples)) + "\n")
        for vi, (p, ra, aa) in enumerate(variants):
            cols = ["|".join(str(int(x)) for x in gts[vi, s * ploidy:(s + 1) * ploidy]) for s in range(n_samples)]
            f.write(f"{chrom}\t{p + 1}\tv{vi}\t{ra.decode()}\t{aa.decode()}\t.\tPASS\t.\tGT\t" + "\t".join(cols) + "\n")


def write_fastq_pair(prefix, block, n_reads, read_len, gz=False):
    """Split an interleaved '\\n'-joined read block (read 2p = mate 1, 2p+1 = mate 2) into two FASTQ files."""
    rec = block.reshape(n_reads, read_len + 1)[:, :read_len]
    qual = b"I" * read_len
    paths = []
    for mate in (0, 1):
        path = f"{prefix}_{mate + 1}.fq" + (".gz" if gz else "")
        opener = gzip.open if gz else open
        with opener(path, "wb") as f:
            rows = rec[mate::2]
            buf = bytearray()
            for i, r in enumerate(rows):
                buf += b"@r%d/%d\n" % (i, mate + 1) + r.tobytes() + b"\n+\n" + qual + b"\n"
            f.write(bytes(buf))
        paths.append(path)
    return paths


def write_fastq_pair_fast(prefix, block, n_reads, read_len, qual="const"):
    """Same files as write_fastq_pair (plain), assembled as one byte matrix per mate instead of a Python loop: record =
    '@r<9 digits>/<mate>' '\n' sequence '\n' '+' '\n' quality '\n' (fixed-width read numbers).  qual: "const" ('I' throughout,
    what the parity tests and the bench use) or "binned" (four quality values drawn 80 / 10 / 7 / 3 %, as a binned
    instrument writes them: the DEFLATE stream of such a file is what a real .fastq.gz looks like to the inflate kernels)."""
    rec = block.reshape(n_reads, read_len + 1)[:, :read_len]
    paths = []
    for mate in (0, 1):
        rows = rec[mate::2]
        n = rows.shape[0]
        width = 2 + 9 + 2 + 1 + read_len + 1 + 2 + read_len + 1
        m = np.empty((n, width), dtype=np.uint8)
        m[:, 0] = ord("@")
        m[:, 1] = ord("r")
        idx = np.arange(n, dtype=np.int64)
        for d in range(9):
            m[:, 2 + d] = (idx // 10 ** (8 - d)) % 10 + ord("0")
        m[:, 11] = ord("/")
        m[:, 12] = ord("1") + mate
        m[:, 13] = 10
        m[:, 14:14 + read_len] = rows
        o = 14 + read_len
        m[:, o] = 10
        m[:, o + 1] = ord("+")
        m[:, o + 2] = 10
        if qual == "binned":
            rng = np.random.default_rng(1234 + mate)
            m[:, o + 3:o + 3 + read_len] = np.frombuffer(b"F:,#", dtype=np.uint8)[rng.choice(4, size=(n, read_len), p=[0.8, 0.1, 0.07, 0.03])]
        else:
            m[:, o + 3:o + 3 + read_len] = ord("I")
        m[:, o + 3 + read_len] = 10
        path = f"{prefix}_{mate + 1}.fq"
        m.tofile(path)
        paths.append(path)
    return paths


def write_fastq_pair_device(prefix, haps, n_pairs, seed, read_len=150, device=0):
    """<prefix>_1.fq / _2.fq with 2 x n_pairs reads of `haps` drawn by the device generator (vgmi_synth_reads_device; mate = read
    parity), assembled as byte matrices: the records of write_fastq_pair_fast with running read numbers.  For workloads of
    tens of millions of reads, where the host generator and a Python loop per record would take minutes."""
    import torch
    L = read_len
    ctx = vgmi.Context(device, buffer_mib=16)
    try:
        off = np.concatenate([[0], np.cumsum([h.size for h in haps])]).astype(np.uint64)
        d_cat = torch.from_numpy(np.concatenate(haps)).cuda(device)
        paths = [f"{prefix}_1.fq", f"{prefix}_2.fq"]
        files = [open(p, "wb") for p in paths]
        chunk = 2_000_000          # reads per piece: 0.3 GB of device text, 0.64 GB of FASTQ per mate on the host
        d_block = torch.empty(chunk * (L + 1), dtype=torch.uint8, device=f"cuda:{device}")
        n_reads = 2 * n_pairs
        for first in range(0, n_reads, chunk):
            n = min(chunk, n_reads - first)
            ctx.synth_reads_device(seed, first, n, L, d_cat, off, d_block)
            torch.cuda.synchronize()
            rec = d_block[: n * (L + 1)].cpu().numpy().reshape(n, L + 1)[:, :L]
            for mate in (0, 1):
                rows = rec[mate::2]
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
                files[mate].write(m.tobytes())
        for f in files:
            f.close()
        return paths
    finally:
        ctx.close()


def sample_haplotypes(ref, variants, gts, sample=0, ploidy=2):
    return [haplotype(ref, variants, gts, sample * ploidy + h) for h in range(ploidy)]


# ---- synthetic graph key sets for large-table performance runs (no reference `construct` involved)
def hash64_np(x, k):
    """include/hash64.hpp:5-14 vectorised (uint64 numpy)."""
    mask = np.uint64((1 << (2 * k)) - 1)
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (~x + (x << np.uint64(21))) & mask
        x = x ^ (x >> np.uint64(24))
        x = ((x + (x << np.uint64(3))) + (x << np.uint64(8))) & mask
        x = x ^ (x >> np.uint64(14))
        x = ((x + (x << np.uint64(2))) + (x << np.uint64(4))) & mask
        x = x ^ (x >> np.uint64(28))
        x = (x + (x << np.uint64(31))) & mask
    return x


def snp_kmer_keys(ref, positions, alts, k=27):
    """Keys (hash64(canonical)<<8|k) of every k-mer that covers a SNP site, for the reference and
    the alternative allele -- what a SNP node of the reference's index holds in essence (about
    2k keys per site).  SNP-only cohorts; sites closer than k share context through the reference."""
    codes = _CODE[ref].astype(np.uint64)
    pos = np.asarray(positions, dtype=np.int64)
    alt = _CODE[np.asarray(alts, dtype=np.uint8)].astype(np.uint64)
    span = 2 * k - 1
    idx = pos[:, None] + np.arange(-(k - 1), k)[None, :]
    ctx = codes[idx]                                  # (n, 2k-1)
    out = []
    for allele in (0, 1):
        c = ctx.copy()
        if allele:
            c[:, k - 1] = alt
        for s in range(k):                            # window c[:, s:s+k]
            w = c[:, s:s + k]
            fwd = np.zeros(len(pos), dtype=np.uint64)
            rc = np.zeros(len(pos), dtype=np.uint64)
            for j in range(k):
                fwd = (fwd << np.uint64(2)) | w[:, j]
                rc = (rc << np.uint64(2)) | (np.uint64(3) - w[:, k - 1 - j])
            canon = np.minimum(fwd, rc)
            out.append((hash64_np(canon, k) << np.uint64(8)) | np.uint64(k))
    return np.unique(np.concatenate(out))


def snp_graph(genome, n_variants, ref_seed=777, var_seed=5, k=27, want_keys=True):
    """Large-table workloads (BASELINE configs 3-5 class) without running `construct`: an iid reference, uniformly
    placed SNPs, the key set of every k-mer covering a site on either allele, and the two haplotypes of the sequenced
    (all-het) sample.  Returns (keys, [ref, alt_haplotype])."""
    ref = make_reference(genome, seed=ref_seed)
    rng = np.random.default_rng(var_seed)
    pos = np.sort(rng.choice(np.arange(100, genome - 100), size=n_variants, replace=False))
    alt_code = (_CODE[ref[pos]] + rng.integers(1, 4, size=n_variants)) % 4
    alts = _ACGT[alt_code]
    # == snp_kmer_keys(ref, pos, alts, k), natively; want_keys=False: a rank that receives the table image by broadcast
    # needs the haplotypes only
    keys = np.unique(vgmi.synth_snp_keys(ref, pos, alts, k)) if want_keys else None
    hap1 = ref.copy()
    hap1[pos] = alts
    return keys, [ref, hap1]


def bgzf_compress_file(src, dst, level=4, block=0xff00):
    """Block-gzip (BGZF, what bgzip / htslib write: SAM spec 4.1) copy of `src`: gzip members of <= 64 KiB whose
    extra field 'BC' carries the member size, closed by the empty EOF block."""
    import struct
    import zlib
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        while True:
            d = fi.read(block)
            if not d:
                break
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            cd = c.compress(d) + c.flush()
            fo.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(cd) + 25) + cd +
                     struct.pack("<II", zlib.crc32(d), len(d)))
        fo.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return dst
