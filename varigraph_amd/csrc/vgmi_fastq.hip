// vgmi_fastq.hip -- FASTQ record parsing on the device (gfx950).
//
// Reference behaviour implemented (file:line under the reference tree):
//   kseq_read                     include/kseq.h:192-232   record = header line, sequence line(s), '+' line, quality
//   FastqKmer::fastq_file_open    src/fastq_kmer.cpp:97-105 `while (kseq_read(ks) >= 0)`: sequence = ks->seq.s,
//                                 mReadBase += ks->seq.l
// for the REGULAR form of a FASTQ file -- every record exactly four lines: '@...', sequence, '+...', quality of the
// sequence's length -- which is what sequencers and every FASTQ writer in use produce.  The kernels check that form
// record by record; the first record that does not have it (multi-line sequence or quality, FASTA record, empty
// sequence, length mismatch, a '\r' or NUL byte anywhere in the chunk, ...) stops the device parser for good and the
// host reader (csrc/host/fastx_reader.cpp, the literal kseq restatement) takes the stream over from that record's first
// byte.  For accepted records kseq_read returns exactly the bytes of line 2:
//   * the previous record left kseq at the byte after its quality line with last_char = 0, so it scans to the next '@'
//     or '>' -- the record's first byte, checked to be '@';
//   * the header line is consumed up to its '\n' whatever it holds;
//   * line 2 starts with a byte other than '\n', '>', '+', '@' (checked), so it is appended whole; line 3 starts with
//     '+' (checked), which ends the sequence;
//   * the rest of line 3 is skipped, line 4 is appended to the quality, and because its length equals the sequence's
//     (checked) the quality loop ends there and the lengths agree;
//   * no '\r' in the chunk (checked): KS_SEP_LINE strips nothing; no NUL (checked): `string(ks->seq.s)` is the whole line.
//
// Data flow per chunk of file text (all on one HIP stream, no host round trip):
//   raw[TAILMAX - tail .. TAILMAX + n)   the tail the previous chunk left (an incomplete record) + the new bytes
//   K1 newline count per 4 KiB tile -> K2 scan -> K3 newline positions -> K4 per-record check and length ->
//   K5 scan of the packed lengths -> K6 copy of every sequence line into the '\n'-joined read block ->
//   K7 bookkeeping (records, bases, consumed bytes, tail) -> K8 tail carried into the other raw buffer ->
//   the count kernels run on the packed block with its length read from device memory (RowParams::n_bytes_dev).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_kernels.h"

namespace vgk {

#define FQ_PIECES 4u           // 16-byte pieces per thread in the byte-parallel kernels, 4 KiB apart
#define FQ_TILE (4096u * FQ_PIECES)   // bytes per workgroup there (256 threads x 16 bytes x FQ_PIECES): a workgroup per 4 KiB was 25 000 workgroups of one
                               // load each per 100 MB chunk -- 0.28 ms for a pass over text that streams in 0.03
#define FQ_NONE 0xFFFFFFFFu

// ---- byte-parallel part -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 fq_load16(const uint8_t* raw, uint32_t off) { return *reinterpret_cast<const uint4*>(raw + off); }

// bit j of the result = byte j of the 16 equals c
__device__ __forceinline__ uint32_t fq_eq_mask(const uint4 v, uint32_t c)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = w[i] ^ (c * 0x01010101u);
        // zero-byte detector, exact per byte: ~(((x & 0x7f7f7f7f) + 0x7f7f7f7f) | x | 0x7f7f7f7f) has bit 7 set iff byte == 0
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
        m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * i);
    }
    return m;
}

__device__ __forceinline__ uint32_t fq_valid_mask(uint32_t off, uint32_t lo, uint32_t hi)   // bit j: lo <= off + j < hi
{
    uint32_t m = 0xFFFFu;
    if (off < lo) m &= lo - off >= 16 ? 0u : (0xFFFFu << (lo - off));
    if (off + 16 > hi) m &= hi <= off ? 0u : (0xFFFFu >> (off + 16 - hi));
    return m & 0xFFFFu;
}

__device__ __forceinline__ uint32_t block_reduce_add(uint32_t v, uint32_t* sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) t += sh[i];
    __syncthreads();
    return t;
}

// exclusive prefix of v over the block's threads (blockDim.x <= 1024); *total = block sum
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* sh, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(inc, o);
        if (lane >= (uint32_t)o) inc += n;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) {
        if (i < wave) base += sh[i];
        tot += sh[i];
    }
    __syncthreads();
    if (total) *total = tot;
    return base + inc - v;
}

// K1: newlines per tile; '\r' / NUL bytes anywhere in the chunk make it dirty
// chunk end: host value, or cut on the device (block-gzip batch with a member that did not inflate)
__device__ __forceinline__ uint32_t fq_end(uint32_t tail_max, uint32_t n_new, const uint32_t* n_new_dev)
{
    if (n_new_dev) {
        const uint32_t d = *n_new_dev;
        n_new = d < n_new ? d : n_new;
    }
    return tail_max + n_new;
}

__global__ __launch_bounds__(256) void fq_count_kernel(const uint8_t* raw, FqState* st, uint32_t tail_max, uint32_t n_new,
                                                       const uint32_t* n_new_dev, uint32_t* tile_nl)
{
    __shared__ uint32_t sh[4];
    const uint32_t end = fq_end(tail_max, n_new, n_new_dev);
    const uint32_t start = st->start, off0 = blockIdx.x * FQ_TILE + threadIdx.x * 16u;
    const bool live = !st->stopped;
    uint32_t n = 0, bad = 0;
    uint4 v[FQ_PIECES];
#pragma unroll
    for (uint32_t j = 0; j < FQ_PIECES; ++j) {      // (the loads first, all in flight)
        const uint32_t off = off0 + j * 4096u;
        v[j] = live && off < end && off + 16 > start ? fq_load16(raw, off) : make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
    }
#pragma unroll
    for (uint32_t j = 0; j < FQ_PIECES; ++j) {
        const uint32_t off = off0 + j * 4096u;
        if (live && off < end && off + 16 > start) {
            const uint32_t ok = fq_valid_mask(off, start, end);
            n += __popc(fq_eq_mask(v[j], '\n') & ok);
            bad |= (fq_eq_mask(v[j], '\r') | fq_eq_mask(v[j], 0)) & ok;
        }
    }
    if (bad) atomicOr(&st->dirty, 1u);
    const uint32_t t = block_reduce_add(n, sh);
    if (threadIdx.x == 0) tile_nl[blockIdx.x] = t;
}

// K2 / K5b: exclusive scan of up to 1024 * per_thread values by one workgroup, in place; the total goes to *total
__global__ __launch_bounds__(1024) void fq_scan_small_kernel(uint32_t* v, uint32_t n, uint32_t* total)
{
    __shared__ uint32_t sh[16];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per, e = b + per < n ? b + per : n;
    uint32_t s = 0, tot;
    if (per <= 16u) {
        // a thread's values in registers, fetched together: one workgroup is the whole launch, and 2 x per loads one after the other
        // (0.22 ms per call, two calls per chunk) were a tenth of a chunk's kernels
        uint32_t x[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) x[i] = b + i < e ? v[b + i] : 0u;
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) s += x[i];
        uint32_t run = block_scan_excl(s, sh, &tot);
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) {
            if (b + i < e) v[b + i] = run;
            run += x[i];
        }
    } else {
        for (uint32_t i = b; i < e; ++i) s += v[i];
        uint32_t run = block_scan_excl(s, sh, &tot);
        for (uint32_t i = b; i < e; ++i) {
            const uint32_t x = v[i];
            v[i] = run;
            run += x;
        }
    }
    if (threadIdx.x == 0 && total) *total = tot;
}

// K3: positions of the newlines, in order
__global__ __launch_bounds__(256) void fq_nlpos_kernel(const uint8_t* raw, FqState* st, uint32_t tail_max, uint32_t n_new,
                                                       const uint32_t* n_new_dev, const uint32_t* tile_base, uint32_t* nlpos,
                                                       uint32_t cap_lines)
{
    __shared__ uint32_t sh[4];
    const uint32_t end = fq_end(tail_max, n_new, n_new_dev);
    if (st->stopped || st->dirty || st->n_lines > cap_lines) return;
    const uint32_t start = st->start, off0 = blockIdx.x * FQ_TILE + threadIdx.x * 16u;
    uint32_t m[FQ_PIECES];
#pragma unroll
    for (uint32_t j = 0; j < FQ_PIECES; ++j) {
        const uint32_t off = off0 + j * 4096u;
        m[j] = off < end && off + 16 > start ? fq_eq_mask(fq_load16(raw, off), '\n') & fq_valid_mask(off, start, end) : 0u;
    }
    uint32_t base = tile_base[blockIdx.x];
#pragma unroll
    for (uint32_t j = 0; j < FQ_PIECES; ++j) {      // piece by piece: the positions of a tile in order
        uint32_t tot, mm = m[j];
        uint32_t pos = base + block_scan_excl(__popc(mm), sh, &tot);
        base += tot;
        while (mm) {
            const uint32_t q = __builtin_ctz(mm);
            mm &= mm - 1;
            nlpos[pos++] = off0 + j * 4096u + q;
        }
    }
}

// ---- record-parallel part -----------------------------------------------------------------------------------------
// K4: one thread per 4-line group: the regular-record checks of the header comment, packed length = sequence + '\n'
__global__ __launch_bounds__(256) void fq_records_kernel(const uint8_t* raw, FqState* st, const uint32_t* nlpos, uint32_t* rec_bytes,
                                                         uint32_t cap_lines)
{
    if (st->stopped || st->dirty || st->n_lines > cap_lines) return;
    const uint32_t R = st->n_lines >> 2;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const uint32_t s0 = r ? nlpos[4 * r - 1] + 1 : st->start;
    const uint32_t e0 = nlpos[4 * r], e1 = nlpos[4 * r + 1], e2 = nlpos[4 * r + 2], e3 = nlpos[4 * r + 3];
    const uint32_t len = e1 - e0 - 1;
    const uint32_t c1 = raw[e0 + 1];
    const bool ok = raw[s0] == '@' && len >= 1 && c1 != '@' && c1 != '+' && c1 != '>' && raw[e1 + 1] == '+' && e3 - e2 - 1 == len;
    rec_bytes[r] = len + 1;
    if (!ok) atomicMin(&st->first_bad, r);
}

// K5a / K5c: scan of rec_bytes over the records: block sums, then (after fq_scan_small_kernel over the sums) the prefix
__global__ __launch_bounds__(1024) void fq_scan_blocks_kernel(const uint32_t* v, const FqState* st, uint32_t* block_sum, uint32_t* out,
                                                             int phase, uint32_t cap_lines)
{
    __shared__ uint32_t sh[16];
    if (st->stopped || st->dirty || st->n_lines > cap_lines) return;   // (uniform) nothing of this chunk is taken; R may exceed the arrays then
    const uint32_t R = st->n_lines >> 2;
    if (blockIdx.x * 1024u >= R) {      // (uniform) the launch is sized for the arrays' capacity, thirteen times a chunk's records: 0.7 ms per call
        if (phase == 0 && threadIdx.x == 0) block_sum[blockIdx.x] = 0;
        return;
    }
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t x = i < R ? v[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_scan_excl(x, sh, &tot);
    if (phase == 0) {
        if (threadIdx.x == 0) block_sum[blockIdx.x] = tot;
    } else if (i < R) {
        out[i] = block_sum[blockIdx.x] + ex;
    }
}

// K7: chunk bookkeeping (one thread) -- runs BEFORE the copy so that fq_pack only touches accepted records
__global__ void fq_finish_kernel(FqState* st, const uint32_t* nlpos, const uint32_t* rec_bytes, const uint32_t* out_off, uint32_t n_new,
                                 const uint32_t* n_new_dev, uint32_t cap_lines, uint32_t tail_max)
{
    if (threadIdx.x || blockIdx.x) return;
    const uint32_t end = fq_end(tail_max, n_new, n_new_dev);
    uint32_t good = 0, consumed_end = st->start, packed = 0;
    if (!st->stopped) {
        if (st->dirty || st->n_lines > cap_lines) {
            st->stopped = 1;     // nothing of this chunk is taken: the host reader resumes at its first byte
        } else {
            const uint32_t R = st->n_lines >> 2;
            good = st->first_bad < R ? st->first_bad : R;
            if (st->first_bad < R) st->stopped = 1;
            if (good) {
                consumed_end = nlpos[4 * good - 1] + 1;
                packed = out_off[good - 1] + rec_bytes[good - 1];
            }
        }
    }
    st->n_good = good;
    st->packed_bytes = packed;
    st->n_records += good;
    st->n_bases += packed - good;
    st->consumed += consumed_end - st->start;
    st->consumed_end = consumed_end;
    uint32_t tail = st->stopped ? 0u : end - consumed_end;
    if (tail > tail_max) {      // a "record" longer than the carry buffer: not a short-read FASTQ
        st->stopped = 1;
        tail = 0;
    }
    st->tail_len = tail;
}

// K6: sequence line of every accepted record -> packed block, one wavefront per record, byte-granular and coalesced
__global__ __launch_bounds__(256) void fq_pack_kernel(const uint8_t* raw, const FqState* st, const uint32_t* nlpos, const uint32_t* out_off,
                                                      uint8_t* packed)
{
    const uint32_t good = st->n_good;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = wave; r < good; r += n_waves) {
        const uint32_t b = nlpos[4 * r] + 1, e = nlpos[4 * r + 1];   // [b, e) = the sequence, raw[e] = '\n'
        uint8_t* dst = packed + out_off[r];
        for (uint32_t i = lane; i <= e - b; i += 64) dst[i] = raw[b + i];
    }
}

// K8: the unconsumed tail goes in front of the next chunk's landing area, and the per-chunk state is re-armed
__global__ __launch_bounds__(256) void fq_carry_kernel(const uint8_t* raw, uint8_t* raw_next, FqState* st, uint32_t tail_max)
{
    const uint32_t tail = st->tail_len, from = st->consumed_end;
    for (uint32_t i = threadIdx.x; i < tail; i += blockDim.x) raw_next[tail_max - tail + i] = raw[from + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        st->start = tail_max - tail;
        st->first_bad = FQ_NONE;
        st->dirty = 0;
        st->n_lines = 0;
    }
}

__global__ void fq_init_kernel(FqState* st, uint32_t tail_max)
{
    if (threadIdx.x || blockIdx.x) return;
    *st = FqState{};
    st->start = tail_max;
    st->first_bad = FQ_NONE;
}

// ---- launcher: everything one chunk needs, in stream order ------------------------------------------------------------
hipError_t launch_fastq_init(FqState* st, uint32_t tail_max, hipStream_t s)
{
    hipLaunchKernelGGL(fq_init_kernel, dim3(1), dim3(1), 0, s, st, tail_max);
    return hipGetLastError();
}

hipError_t launch_fastq_chunk(const FqBuffers& b, uint32_t n_new, hipStream_t s, const uint32_t* n_new_dev)
{
    const uint32_t end = b.tail_max + n_new;
    const uint32_t n_tiles = (end + FQ_TILE - 1) / FQ_TILE;
    const uint32_t cap_rec = b.cap_lines / 4;
    const uint32_t n_rblk = (cap_rec + 1023u) / 1024u;
    hipLaunchKernelGGL(fq_count_kernel, dim3(n_tiles), dim3(256), 0, s, b.raw, b.state, b.tail_max, n_new, n_new_dev, b.tile);
    hipLaunchKernelGGL(fq_scan_small_kernel, dim3(1), dim3(1024), 0, s, b.tile, n_tiles, &b.state->n_lines);
    hipLaunchKernelGGL(fq_nlpos_kernel, dim3(n_tiles), dim3(256), 0, s, b.raw, b.state, b.tail_max, n_new, n_new_dev, b.tile, b.nlpos, b.cap_lines);
    hipLaunchKernelGGL(fq_records_kernel, dim3((cap_rec + 255u) / 256u), dim3(256), 0, s, b.raw, b.state, b.nlpos, b.rec_bytes, b.cap_lines);
    hipLaunchKernelGGL(fq_scan_blocks_kernel, dim3(n_rblk), dim3(1024), 0, s, b.rec_bytes, b.state, b.block_sum, b.out_off, 0, b.cap_lines);
    hipLaunchKernelGGL(fq_scan_small_kernel, dim3(1), dim3(1024), 0, s, b.block_sum, n_rblk, (uint32_t*)nullptr);
    hipLaunchKernelGGL(fq_scan_blocks_kernel, dim3(n_rblk), dim3(1024), 0, s, b.rec_bytes, b.state, b.block_sum, b.out_off, 1, b.cap_lines);
    hipLaunchKernelGGL(fq_finish_kernel, dim3(1), dim3(1), 0, s, b.state, b.nlpos, b.rec_bytes, b.out_off, n_new, n_new_dev, b.cap_lines, b.tail_max);
    hipLaunchKernelGGL(fq_pack_kernel, dim3(2048), dim3(256), 0, s, b.raw, b.state, b.nlpos, b.out_off, b.packed);
    hipLaunchKernelGGL(fq_carry_kernel, dim3(1), dim3(256), 0, s, b.raw, b.raw_next, b.state, b.tail_max);
    return hipGetLastError();
}

}  // namespace vgk
