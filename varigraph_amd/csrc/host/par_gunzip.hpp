// par_gunzip.hpp -- ONE ordinary gzip stream decoded by several threads (the reference reads FASTQ
// through zlib's gzread, include/kseq.h:59-72 with src/fastq_kmer.cpp:74-78, one inflate per file).
//
// A DEFLATE stream has no index, and a block can refer 32 KiB back, so a thread cannot simply start in the middle.  It can
// however (the approach of pugz / rapidgzip, restated here):
//   1. FIND a block start by trying bit positions: a dynamic-code block header is a few hundred bits that must describe
//      three complete prefix codes, and the blocks behind it must decode -- a false start does not survive that;
//   2. DECODE from there with the 32 KiB in front unknown: output symbols are 16 bits wide, a copy that reaches into the
//      unknown window yields MARKERS (0x8000 + window offset) that later copies carry along like any other symbol;
//   3. RESOLVE the markers once the thread in front has finished: its last 32 KiB are the window.  The end of a span is
//      resolved first (it is the next span's window), so the chain through the file is 32 KiB per span, the rest runs
//      side by side.
// Spans are checked against each other (a span must start at the very bit the one in front ended on); a span that cannot
// be trusted -- a false start, a stored or fixed-code block at the seam, a member boundary, damage -- is decoded again
// from the known position with the known window, in order.  So the bytes are those of a serial decoder in every case,
// including where a damaged or truncated stream ends: member CRC-32 and ISIZE are checked, concatenated members are
// followed, bytes after the last member that are not a gzip header are ignored (gzread's rules, byte_source.hpp).
#pragma once
#include <memory>
#include <string>

#include "byte_source.hpp"

namespace vgh {

// nullptr when the file cannot be mapped (the caller falls back to the serial decoder).  span_bytes: compressed bytes per
// span (0: default; tests use small spans to get many seams out of small files).
std::unique_ptr<ByteSource> open_parallel_gunzip(const std::string& path, unsigned threads, size_t span_bytes = 0);

}  // namespace vgh
