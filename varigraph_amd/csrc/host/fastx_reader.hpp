// fastx_reader.hpp -- FASTA/FASTQ(.gz) record reader with the semantics of klib's kseq_read as
// the reference uses it (include/kseq.h:192-232 through FastqKmer::fastq_file_open,
// src/fastq_kmer.cpp:74-105):
//   * records start at the next '>' or '@'; the name ends at the first whitespace
//   * sequence = concatenation of the following lines up to a line starting with '>', '+' or '@'
//     (empty lines skipped, a trailing '\r' dropped per line when the sequence is longer than 1)
//   * FASTQ: the rest of the '+' line is skipped, quality lines are appended until they are at
//     least as long as the sequence; a length mismatch or a missing quality is a truncated record
//     (-2) and the reference's `while (kseq_read(ks) >= 0)` loop stops reading the file there
//   * the file is opened with gzopen, so plain and gzip inputs both work
#pragma once
#include <zlib.h>

#include <cstdint>
#include <string>

namespace vgh {

class FastxReader {
public:
    explicit FastxReader(const std::string& path);  // throws std::runtime_error if it cannot open
    ~FastxReader();
    FastxReader(const FastxReader&) = delete;
    FastxReader& operator=(const FastxReader&) = delete;

    // >= 0: sequence length (seq() holds it); -1: end of file; -2: truncated quality
    long next();
    const std::string& seq() const { return seq_; }
    const std::string& name() const { return name_; }  // up to the first whitespace of the header line

private:
    int getc();
    // append up to (not including) the next '\n' to s; returns false at EOF with nothing read
    bool get_line(std::string& s, bool append);

    gzFile fp_;
    unsigned char* buf_;
    int begin_ = 0, end_ = 0;
    bool eof_ = false;
    int last_char_ = 0;
    std::string seq_, qual_, name_;
};

}  // namespace vgh
