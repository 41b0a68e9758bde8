// fastx_reader.hpp -- FASTA/FASTQ(.gz) record reader with the semantics of klib's kseq_read as
// the reference uses it (include/kseq.h:192-232 through FastqKmer::fastq_file_open,
// src/fastq_kmer.cpp:74-105):
//   * records start at the next '>' or '@'; the name ends at the first whitespace
//   * sequence = concatenation of the following lines up to a line starting with '>', '+' or '@'
//     (empty lines skipped, a trailing '\r' dropped per line when the sequence is longer than 1)
//   * FASTQ: the rest of the '+' line is skipped, quality lines are appended until they are at
//     least as long as the sequence; a length mismatch or a missing quality is a truncated record
//     (-2) and the reference's `while (kseq_read(ks) >= 0)` loop stops reading the file there
//   * the bytes come from a ByteSource (plain, gzip or block-gzip input, as gzopen would deliver them) whose
//     inflate runs on its own thread(s), `decode_threads` of them for block gzip
#pragma once
#include <cstdint>
#include <memory>
#include <string>

#include "byte_source.hpp"

namespace vgh {

class FastxReader {
public:
    explicit FastxReader(const std::string& path, unsigned decode_threads = 1);  // throws std::runtime_error if it cannot open
    explicit FastxReader(std::unique_ptr<ByteSource> src);   // any byte stream, read from a record boundary on
    ~FastxReader();
    FastxReader(const FastxReader&) = delete;
    FastxReader& operator=(const FastxReader&) = delete;

    // >= 0: sequence length (seq() holds it); -1: end of file; -2: truncated quality
    long next();
    const std::string& seq() const { return seq_; }
    const std::string& name() const { return name_; }  // up to the first whitespace of the header line
    const char* source_kind() const { return src_->kind(); }

private:
    int getc();
    // append up to (not including) the next '\n' to s; returns false at EOF with nothing read
    bool get_line(std::string& s, bool append);

    bool refill();   // false at the end of the data
    bool skip_line();

    std::unique_ptr<ByteSource> src_;
    const unsigned char* cur_ = nullptr;
    const unsigned char* end_ = nullptr;
    bool eof_ = false;
    int last_char_ = 0;
    std::string seq_, qual_, name_;
};

}  // namespace vgh
