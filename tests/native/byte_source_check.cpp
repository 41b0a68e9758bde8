// byte_source_check.cpp -- the byte-source combinators the device-side FASTQ path hands a stream back to the host
// reader with (csrc/host/byte_source.hpp: skip, concat, from_memory, open_at) and FastxReader over them.
// usage: byte_source_check <file> <mode> <a> [<b>]
//   cat    <file>            : every byte of the source as opened (plain / gzip / block gzip)            -> stdout
//   skip   <file> <n>        : the same without its first n bytes
//   at     <file> <offset>   : ByteSource::open_at (a member boundary of a compressed file, any offset of a plain one)
//   glue   <file> <n> <m>    : concat(from_memory(first n bytes of the decoded stream), skip(open, m))
//   seqs   <file> <n>        : sequences FastxReader yields from skip(open, n), one per line
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "byte_source.hpp"
#include "fastx_reader.hpp"

using namespace vgh;

static std::string slurp(ByteSource& s)
{
    std::string out;
    const unsigned char* p;
    size_t n;
    while (s.next_chunk(p, n)) out.append(reinterpret_cast<const char*>(p), n);
    return out;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string path = argv[1], mode = argv[2];
    const unsigned long long a = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0, b = argc > 4 ? strtoull(argv[4], nullptr, 10) : 0;
    try {
        std::string out;
        if (mode == "cat") {
            out = slurp(*ByteSource::open(path, 3));
        } else if (mode == "skip") {
            out = slurp(*ByteSource::skip(ByteSource::open(path, 3), a));
        } else if (mode == "at") {
            out = slurp(*ByteSource::open_at(path, a, 3));
        } else if (mode == "glue") {
            const std::string all = slurp(*ByteSource::open(path, 2));
            const std::string head = all.substr(0, (size_t)a);
            out = slurp(*ByteSource::concat(ByteSource::from_memory(head.data(), head.size()), ByteSource::skip(ByteSource::open(path, 2), b)));
        } else if (mode == "seqs") {
            FastxReader rd(ByteSource::skip(ByteSource::open(path, 2), a));
            while (rd.next() >= 0) {
                out += rd.seq();
                out += '\n';
            }
        } else {
            return 2;
        }
        fwrite(out.data(), 1, out.size(), stdout);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
