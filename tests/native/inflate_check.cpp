// Differential check of csrc/host/fast_inflate.cpp against zlib on the files named on the command line (test
// infrastructure, built with -fsanitize=address,undefined by tests/test_fast_inflate_cpu.py).  For every file the
// decoder runs with tiny (4 KiB) and large output buffers.  Clean streams must give zlib's bytes and end Clean;
// truncated streams must give exactly the bytes zlib delivers before it runs out of input; damaged streams must give
// a prefix of zlib's output (and never crash).  Prints "<files> files, <n> mismatches".
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fast_inflate.hpp"

using vgh::GunzipEnd;
using vgh::GunzipIO;

static std::vector<unsigned char> slurp(const char* p)
{
    std::vector<unsigned char> v;
    FILE* f = fopen(p, "rb");
    if (!f) return v;
    unsigned char b[65536];
    size_t n;
    while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n);
    fclose(f);
    return v;
}

// zlib over concatenated members the way gzread walks them; status 0 clean, 1 ran out of input, 2 data error
static std::vector<unsigned char> zlib_decode(const std::vector<unsigned char>& in, int& status)
{
    std::vector<unsigned char> out, buf(1 << 16);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, 15 + 16);
    zs.next_in = const_cast<Bytef*>(in.data());
    zs.avail_in = (uInt)in.size();
    status = 0;
    for (;;) {
        zs.next_out = buf.data();
        zs.avail_out = (uInt)buf.size();
        const int r = inflate(&zs, Z_NO_FLUSH);
        out.insert(out.end(), buf.data(), buf.data() + (buf.size() - zs.avail_out));
        if (r == Z_STREAM_END) {
            if (zs.avail_in < 2 || zs.next_in[0] != 0x1f || zs.next_in[1] != 0x8b) break;
            inflateReset(&zs);
            continue;
        }
        if ((r == Z_BUF_ERROR || r == Z_OK) && zs.avail_in == 0 && zs.avail_out != 0) { status = 1; break; }
        if (r != Z_OK && r != Z_BUF_ERROR) { status = 2; break; }
    }
    inflateEnd(&zs);
    return out;
}

int main(int argc, char** argv)
{
    int bad = 0, files = 0;
    for (int a = 1; a < argc; ++a) {
        const auto in = slurp(argv[a]);
        if (in.size() < 2 || in[0] != 0x1f || in[1] != 0x8b) continue;
        ++files;
        int zst;
        const auto want = zlib_decode(in, zst);
        for (size_t cap : {(size_t)4096, (size_t)1 << 20}) {
            std::vector<unsigned char> got, buf(32768 + cap + 512);
            size_t pos = 0;
            GunzipIO io;
            io.read = [&](unsigned char* d, size_t n) {
                const size_t m = std::min(n, in.size() - pos);
                memcpy(d, in.data() + pos, m);
                pos += m;
                return m;
            };
            io.next_buffer = [&](size_t h, size_t) { return buf.data() + h; };
            io.commit = [&](size_t n) { got.insert(got.end(), buf.data() + 32768, buf.data() + 32768 + n); };
            const GunzipEnd e = vgh::fast_gunzip(io, cap);
            bool ok;
            if (zst == 0) ok = e == GunzipEnd::Clean && got == want;
            else if (zst == 1) ok = e != GunzipEnd::Clean && got == want;
            else ok = e != GunzipEnd::Clean && got.size() <= want.size() && memcmp(got.data(), want.data(), got.size()) == 0;
            if (!ok) {
                ++bad;
                printf("MISMATCH %s cap=%zu zlib=%d end=%d got=%zu want=%zu\n", argv[a], cap, zst, (int)e, got.size(), want.size());
            }
        }
    }
    printf("%d files, %d mismatches\n", files, bad);
    return bad != 0;
}
