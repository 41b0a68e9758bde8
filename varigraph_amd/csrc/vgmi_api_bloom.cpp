// vgmi_api_bloom.cpp -- the construct side's counting Bloom filter (vgmi_bloom_*): K3 update, K4 query, the reference's file format
#include "vgmi_ctx.h"

extern "C" {

/* ---------------------------------------------------------------- Bloom */

int vgmi_bloom_params(uint64_t n, double p, uint64_t* m, uint32_t* n_hash)
{
    // BloomFilter::_calculate_size / _calculate_num_hashes (src/counting_bloom_filter.cpp:70-77)
    const uint64_t mm = (uint64_t)std::ceil(((double)n * std::log(p)) / std::log(1.0 / std::pow(2.0, std::log(2.0))));
    if (m) *m = mm;
    if (n_hash) *n_hash = (uint32_t)std::round((double)mm * std::log(2.0) / (double)n);
    return VGMI_OK;
}

int vgmi_bloom_create(vgmi_ctx* c, uint64_t m, uint32_t n_hash, const uint64_t* seeds)
{
    if (!c || !seeds) return VGMI_E_INVALID;
    if (m == 0 || n_hash == 0 || n_hash > VG_BLOOM_MAX_HASH) return fail(c, VGMI_E_INVALID, "bad Bloom geometry");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->bv.filter) (void)hipFree(c->bv.filter);
    if (c->bb_scratch) (void)hipFree(c->bb_scratch);
    c->bb_scratch = nullptr;
    c->bb_cap = 0;
    c->bv = BloomView{};
    c->has_bloom = false;
    c->bloom_alloc = ((m + 3) & ~3ULL) + 16;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->bv.filter), c->bloom_alloc));
    HIPCHK(c, hipMemset(c->bv.filter, 0, c->bloom_alloc));
    c->bv.m = m;
    c->bv.magic = UINT64_MAX / m;
    c->bv.n_hash = n_hash;
    for (uint32_t i = 0; i < n_hash; ++i) c->bv.seeds[i] = (uint32_t)seeds[i];  // `unsigned int seed`
    for (uint32_t i = 0; i < n_hash; ++i) c->bloom_seeds64[i] = seeds[i];
    c->has_bloom = true;
    return VGMI_OK;
}

int vgmi_bloom_add_seq_device(vgmi_ctx* c, const char* dev_bases, uint64_t len, uint32_t k)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    if (k < 1 || k > 28) return fail(c, VGMI_E_INVALID, "k must be in 1..28");
    if (len == 0) return fail(c, VGMI_E_EMPTY_READ, "empty sequence (reference: assert(len > 0), kmer.cpp:27)");
    if (reinterpret_cast<uintptr_t>(dev_bases) & 15) return fail(c, VGMI_E_INVALID, "dev_bases must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    RowParams p = row_params(c, dev_bases, len, k);
    if (k & 1) {
        uint32_t grid, block;
        rows_geometry(c, false, grid, block);
        // long sequences: positions binned by 128 KiB chunk of the filter and counted in LDS (vgmi_bloom_bin.hip) -- worth it when
        // every chunk gets a few thousand positions; VGMI_BLOOM_BINNED=0 keeps the direct form
        static const bool binned = !(getenv("VGMI_BLOOM_BINNED") && getenv("VGMI_BLOOM_BINNED")[0] == '0');
        const BloomBinPlan plan = binned && len >= (4u << 20) ? bloom_bin_plan(c->bv.m, c->bv.n_hash, len) : BloomBinPlan{};
        if (plan.ok && (double)len * c->bv.n_hash >= 2048.0 * plan.n_chunks) {
            const size_t keys_bytes = (len * 8 + 255) & ~(size_t)255, need = keys_bytes + plan.scratch_bytes;
            if (c->bb_cap < need) {
                if (c->bb_scratch) (void)hipFree(c->bb_scratch);
                c->bb_scratch = nullptr;
                c->bb_cap = 0;
                if (hipMalloc(reinterpret_cast<void**>(&c->bb_scratch), need) == hipSuccess) c->bb_cap = need;
                else (void)hipGetLastError();          // no room: the direct form
            }
            if (c->bb_cap >= need) {
                RowParams pk = p;
                pk.keys_out = reinterpret_cast<uint64_t*>(c->bb_scratch);
                HIPCHK(c, launch_rows(K_MODE_KEYS, false, pk, grid, block, c->stream));
                int overflowed = 0;
                HIPCHK(c, launch_bloom_binned(c->bv, pk.keys_out, len, plan, c->bb_scratch + keys_bytes, c->n_cu, c->stream, &overflowed));
                if (!overflowed) return VGMI_OK;       // (a bin out of room -- one k-mer repeated through the call: nothing applied, the direct form does it)
            }
        }
        HIPCHK(c, launch_rows(K_MODE_BLOOM, false, p, grid, block, c->stream));
    } else {
        // even k: the sequential state machine, one lane per 1 KiB segment with its state rebuilt by look-back
        HIPCHK(c, launch_bloom_even(p, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VGMI_OK;
}

int vgmi_bloom_add_seq(vgmi_ctx* c, const char* bases, uint64_t len, uint32_t k)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    if (len == 0) return fail(c, VGMI_E_EMPTY_READ, "empty sequence (reference: assert(len > 0), kmer.cpp:27)");
    if (!bases) return fail(c, VGMI_E_INVALID, "bases is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    char* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), len + 16));
    hipError_t e = hipMemcpy(d, bases, len, hipMemcpyHostToDevice);
    int rc = VGMI_OK;
    if (e == hipSuccess) {
        rc = vgmi_bloom_add_seq_device(c, d, len, k);
        if (rc == VGMI_OK) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d);
    if (rc) return rc;
    HIPCHK(c, e);
    return VGMI_OK;
}

// BloomFilter::save / load (src/counting_bloom_filter.cpp:126-190): u64 size | u32 numHashes | numHashes x u64 seed | size bytes
int vgmi_bloom_save_file(vgmi_ctx* c, const char* path)
{
    if (!c || !path) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    std::vector<uint8_t> filt(c->bv.m);
    int rc = vgmi_bloom_fetch(c, filt.data());
    if (rc != VGMI_OK) return rc;
    FILE* fp = fopen(path, "wb");
    if (!fp) return fail(c, VGMI_E_INVALID, std::string("'") + path + "': No such file or directory.");
    const uint64_t m = c->bv.m;
    const uint32_t nh = c->bv.n_hash;
    bool ok = fwrite(&m, 8, 1, fp) == 1 && fwrite(&nh, 4, 1, fp) == 1 && fwrite(c->bloom_seeds64, 8, nh, fp) == nh &&
              fwrite(filt.data(), 1, filt.size(), fp) == filt.size();
    if (fclose(fp) != 0) ok = false;
    return ok ? VGMI_OK : fail(c, VGMI_E_INVALID, std::string("'") + path + "': write error.");
}

int vgmi_bloom_load_file(vgmi_ctx* c, const char* path)
{
    if (!c || !path) return VGMI_E_INVALID;
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(c, VGMI_E_INVALID, std::string("'") + path + "': No such file or directory.");
    uint64_t m = 0, seeds[VG_BLOOM_MAX_HASH];
    uint32_t nh = 0;
    bool ok = fread(&m, 8, 1, fp) == 1 && fread(&nh, 4, 1, fp) == 1 && nh >= 1 && nh <= VG_BLOOM_MAX_HASH && m > 0 &&
              fread(seeds, 8, nh, fp) == nh;
    std::vector<uint8_t> filt;
    if (ok) {
        const long at = ftell(fp);
        ok = at >= 0 && fseek(fp, 0, SEEK_END) == 0 && (uint64_t)(ftell(fp) - at) == m && fseek(fp, at, SEEK_SET) == 0;   // sized by the file, not by its header
        if (ok) {
            filt.resize(m);
            ok = fread(filt.data(), 1, filt.size(), fp) == filt.size();
        }
    }
    fclose(fp);
    if (!ok) return fail(c, VGMI_E_INVALID, std::string("'") + path + "': not a counting Bloom filter file.");
    int rc = vgmi_bloom_create(c, m, nh, seeds);
    if (rc != VGMI_OK) return rc;
    return vgmi_bloom_load(c, filt.data());
}

int vgmi_bloom_fetch(vgmi_ctx* c, uint8_t* out)
{
    if (!c || !out) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->bv.filter, c->bv.m, hipMemcpyDeviceToHost));
    return VGMI_OK;
}

int vgmi_bloom_load(vgmi_ctx* c, const uint8_t* in)
{
    if (!c || !in) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->bv.filter, in, c->bv.m, hipMemcpyHostToDevice));
    return VGMI_OK;
}

int vgmi_bloom_query(vgmi_ctx* c, const uint64_t* keys, size_t n, uint8_t* min_out, uint8_t* nz_out)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    if (n == 0) return VGMI_OK;
    if (!keys) return fail(c, VGMI_E_INVALID, "keys is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t* d_k = nullptr;
    uint8_t* d_o = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_k), n * 8));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_o), 2 * n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_k, keys, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_bloom_query(c->bv, d_k, n, d_o, d_o + n, c->stream);
    if (e == hipSuccess && min_out) e = hipMemcpyAsync(min_out, d_o, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && nz_out) e = hipMemcpyAsync(nz_out, d_o + n, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_k);
    if (d_o) (void)hipFree(d_o);
    HIPCHK(c, e);
    return VGMI_OK;
}

}  // extern "C"
