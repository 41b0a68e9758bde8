// byte_source.hpp -- the bytes of an input file as zlib's gzopen/gzread would deliver them (what the reference
// reads through: kseq over gzFile, include/kseq.h:59-72 with src/fastq_kmer.cpp:74-78; GzChunkReader for VCFs), with
// the inflate work taken off the parsing thread:
//   plain file            read(2) in 256 KiB chunks (gzopen's transparent mode)
//   gzip stream           one decode thread per file running this repo's inflate (fast_inflate.hpp, about twice zlib's
//                         rate) ahead of the parser; concatenated members are followed, bytes after the last member
//                         that are not a gzip header are ignored (gz_look)
//   block gzip (BGZF)     members carrying the 'BC' extra field (bgzip, htslib): the block sizes are in the headers,
//                         so `decode_threads` workers inflate blocks side by side and the chunks come back in order
// A damaged stream delivers what decoded before the damage and then ends, like the reference's
// `while (kseq_read(ks) >= 0)` loop does when gzread returns -1.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

namespace vgh {

class ByteSource {
public:
    virtual ~ByteSource() = default;
    // next run of bytes (owned by the source, valid until the next call); false at the end of the data
    virtual bool next_chunk(const unsigned char*& p, size_t& n) = 0;
    virtual const char* kind() const = 0;   // "plain" | "gzip" | "bgzf"

    // throws std::runtime_error("'<path>': No such file or directory.") like the callers' gzopen checks
    static std::unique_ptr<ByteSource> open(const std::string& path, unsigned decode_threads = 1);
    // the same from a byte offset of the file on (a member boundary of a compressed file)
    static std::unique_ptr<ByteSource> open_at(const std::string& path, uint64_t offset, unsigned decode_threads = 1);
    // `a` to its end, then `b`
    static std::unique_ptr<ByteSource> concat(std::unique_ptr<ByteSource> a, std::unique_ptr<ByteSource> b);
    // a run of bytes already in memory (not owned; must outlive the source)
    static std::unique_ptr<ByteSource> from_memory(const void* data, size_t n);
    // `inner` without its first `n` bytes (the device-side FASTQ parser hands a stream over at a byte offset)
    static std::unique_ptr<ByteSource> skip(std::unique_ptr<ByteSource> inner, uint64_t n);
};

}  // namespace vgh
