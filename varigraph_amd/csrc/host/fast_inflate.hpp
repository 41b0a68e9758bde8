// fast_inflate.hpp -- streaming gunzip for the ingest path: DEFLATE (RFC 1951) inside gzip members
// (RFC 1952) decoded with a 64-bit bit buffer refilled eight bytes at a time, 11-bit / 8-bit first-level Huffman
// tables whose entries carry the decoded base value and extra-bit count, 16-byte match copies, and a carry-less-multiply
// CRC-32.  The reference reads its FASTQ through zlib's gzread (include/kseq.h:59-72, src/fastq_kmer.cpp:74-78), which
// bounds its ingest at one inflate per file on the main thread; this decoder delivers the same bytes about twice as fast
// on a thread of its own (byte_source.cpp).  Same stream semantics as gzread: concatenated members are followed, bytes
// after the last member that are not a gzip header are ignored, a damaged or truncated stream ends the data after what
// decoded cleanly.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

namespace vgh {

struct GunzipIO {
    // more compressed bytes: up to n into dst, 0 at the end of the file
    std::function<size_t(unsigned char* dst, size_t n)> read;
    // an output buffer of `capacity` writable bytes preceded by `history` writable bytes (the decoder keeps the last
    // 32 KiB of output there); nullptr stops the decoder
    std::function<unsigned char*(size_t history, size_t capacity)> next_buffer;
    // the buffer handed out last holds n decoded bytes at its start
    std::function<void(size_t n)> commit;
};

enum class GunzipEnd { Clean, Truncated, Corrupt, Stopped };

// decodes every member from the current read position; the first two bytes must be the gzip magic
GunzipEnd fast_gunzip(const GunzipIO& io, size_t buffer_capacity);

// CRC-32 (gzip polynomial) continued over n bytes; carry-less multiply when the CPU has it
uint32_t crc32_fast(uint32_t crc, const unsigned char* p, size_t n);

}  // namespace vgh
