// Differential check of csrc/host/par_gunzip.cpp (one gzip stream, several decoding threads) against the serial decoder
// of csrc/host/fast_inflate.cpp (itself checked against zlib by inflate_check.cpp) on the files named on the command line
// (test infrastructure, built with -fsanitize=address,undefined by tests/test_par_gunzip_cpu.py).  Every file is decoded
// with several (threads, span) pairs -- spans of 1 KiB put a seam into nearly every DEFLATE block -- and must deliver
// exactly the serial decoder's bytes: clean, truncated or damaged alike.  Prints "<files> files, <n> mismatches".
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fast_inflate.hpp"
#include "par_gunzip.hpp"

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

int main(int argc, char** argv)
{
    int bad = 0, files = 0;
    for (int a = 1; a < argc; ++a) {
        const auto in = slurp(argv[a]);
        if (in.size() < 18 || in[0] != 0x1f || in[1] != 0x8b) continue;
        ++files;
        std::vector<unsigned char> want, buf(32768 + (1 << 20) + 512);
        {
            size_t pos = 0;
            GunzipIO io;
            io.read = [&](unsigned char* d, size_t n) {
                const size_t m = std::min(n, in.size() - pos);
                memcpy(d, in.data() + pos, m);
                pos += m;
                return m;
            };
            io.next_buffer = [&](size_t h, size_t) { return buf.data() + h; };
            io.commit = [&](size_t n) { want.insert(want.end(), buf.data() + 32768, buf.data() + 32768 + n); };
            (void)vgh::fast_gunzip(io, (size_t)1 << 20);
        }
        const struct { unsigned threads; size_t span; } cfg[] = {{2, 1024}, {3, 4096}, {8, 1024}, {4, 65536}, {2, 0}};
        for (const auto& c : cfg) {
            auto src = vgh::open_parallel_gunzip(argv[a], c.threads, c.span);
            if (!src) {
                ++bad;
                printf("MISMATCH %s: not opened\n", argv[a]);
                continue;
            }
            std::vector<unsigned char> got;
            const unsigned char* p = nullptr;
            size_t n = 0;
            while (src->next_chunk(p, n)) got.insert(got.end(), p, p + n);
            if (got != want) {
                ++bad;
                size_t d = 0;
                while (d < got.size() && d < want.size() && got[d] == want[d]) ++d;
                printf("MISMATCH %s threads=%u span=%zu got=%zu want=%zu first difference at %zu\n", argv[a], c.threads, c.span, got.size(),
                       want.size(), d);
            }
        }
    }
    printf("%d files, %d mismatches\n", files, bad);
    return bad != 0;
}
