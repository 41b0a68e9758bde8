#include "fastx_reader.hpp"

#include <cstring>
#include <stdexcept>

namespace vgh {

FastxReader::FastxReader(const std::string& path, unsigned decode_threads) : src_(ByteSource::open(path, decode_threads)) {}

FastxReader::FastxReader(std::unique_ptr<ByteSource> src) : src_(std::move(src)) {}

FastxReader::~FastxReader() = default;

bool FastxReader::refill()
{
    if (eof_) return false;
    size_t n = 0;
    if (!src_->next_chunk(cur_, n) || n == 0) {
        eof_ = true;
        cur_ = end_ = nullptr;
        return false;
    }
    end_ = cur_ + n;
    return true;
}

int FastxReader::getc()
{
    if (cur_ >= end_ && !refill()) return -1;
    return *cur_++;
}

bool FastxReader::get_line(std::string& s, bool append)
{
    if (!append) s.clear();
    if (cur_ >= end_ && !refill()) return false;
    for (;;) {
        const unsigned char* nl = static_cast<const unsigned char*>(memchr(cur_, '\n', (size_t)(end_ - cur_)));
        const unsigned char* stop = nl ? nl : end_;
        s.append(reinterpret_cast<const char*>(cur_), (size_t)(stop - cur_));
        cur_ = nl ? nl + 1 : end_;
        if (nl || !refill()) break;
    }
    if (s.size() > 1 && s.back() == '\r') s.pop_back();  // kseq.h: KS_SEP_LINE strips one trailing '\r'
    return true;
}

// drops the rest of the current line; false if the data ends before a '\n'
bool FastxReader::skip_line()
{
    for (;;) {
        if (cur_ >= end_ && !refill()) return false;
        const unsigned char* nl = static_cast<const unsigned char*>(memchr(cur_, '\n', (size_t)(end_ - cur_)));
        if (nl) {
            cur_ = nl + 1;
            return true;
        }
        cur_ = end_;
    }
}

long FastxReader::next()
{
    int c;
    if (last_char_ == 0) {  // jump to the next header line
        while ((c = getc()) != -1 && c != '>' && c != '@') {}
        if (c == -1) return -1;
        last_char_ = c;
    }
    seq_.clear();
    qual_.clear();
    // name up to the first whitespace, then the rest of the header line (comment)
    {
        bool any = false;
        name_.clear();
        c = -1;
        for (;;) {   // whole runs of name characters at a time
            if (cur_ >= end_ && !refill()) break;
            any = true;
            const unsigned char* p = cur_;
            while (p < end_ && !(*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;   // isspace: ' ', \t \n \v \f \r
            name_.append(reinterpret_cast<const char*>(cur_), (size_t)(p - cur_));
            cur_ = p;
            if (p < end_) {
                c = *cur_++;
                break;
            }
        }
        if (!any) return -1;  // EOF right after the header character
        if (c != '\n' && c != -1) skip_line();
    }
    while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        seq_.push_back((char)c);
        get_line(seq_, true);
    }
    if (c == '>' || c == '@') last_char_ = c;
    if (c != '+') return (long)seq_.size();  // FASTA
    if (!skip_line()) return -2;   // rest of the '+' line
    while (get_line(qual_, true) && qual_.size() < seq_.size()) {}
    last_char_ = 0;
    if (seq_.size() != qual_.size()) return -2;
    return (long)seq_.size();
}

}  // namespace vgh
