#include "fastx_reader.hpp"

#include <cstring>
#include <stdexcept>

namespace vgh {

static const int kBufSize = 1 << 18;

FastxReader::FastxReader(const std::string& path)
{
    fp_ = gzopen(path.c_str(), "rb");
    if (!fp_) throw std::runtime_error("'" + path + "': No such file or directory.");
    gzbuffer(fp_, 1 << 20);
    buf_ = new unsigned char[kBufSize];
}

FastxReader::~FastxReader()
{
    if (fp_) gzclose(fp_);
    delete[] buf_;
}

int FastxReader::getc()
{
    if (begin_ >= end_) {
        if (eof_) return -1;
        begin_ = 0;
        end_ = gzread(fp_, buf_, kBufSize);
        if (end_ < kBufSize) eof_ = true;
        if (end_ <= 0) { end_ = 0; return -1; }
    }
    return buf_[begin_++];
}

bool FastxReader::get_line(std::string& s, bool append)
{
    if (!append) s.clear();
    if (begin_ >= end_ && eof_) return false;
    for (;;) {
        if (begin_ >= end_) {
            if (eof_) break;
            begin_ = 0;
            end_ = gzread(fp_, buf_, kBufSize);
            if (end_ < kBufSize) eof_ = true;
            if (end_ <= 0) { end_ = 0; break; }
        }
        const unsigned char* nl = static_cast<const unsigned char*>(memchr(buf_ + begin_, '\n', (size_t)(end_ - begin_)));
        const int i = nl ? (int)(nl - buf_) : end_;
        s.append(reinterpret_cast<const char*>(buf_ + begin_), (size_t)(i - begin_));
        begin_ = i + 1;
        if (nl) break;
    }
    if (s.size() > 1 && s.back() == '\r') s.pop_back();  // kseq.h: KS_SEP_LINE strips one trailing '\r'
    return true;
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
        for (;;) {
            c = getc();
            if (c == -1) break;
            any = true;
            if (c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r') break;
            name_.push_back((char)c);
        }
        if (!any) return -1;  // EOF right after the header character
        if (c != '\n' && c != -1) {
            std::string comment;
            get_line(comment, false);
        }
    }
    while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        seq_.push_back((char)c);
        get_line(seq_, true);
    }
    if (c == '>' || c == '@') last_char_ = c;
    if (c != '+') return (long)seq_.size();  // FASTA
    while ((c = getc()) != -1 && c != '\n') {}
    if (c == -1) return -2;
    while (get_line(qual_, true) && qual_.size() < seq_.size()) {}
    last_char_ = 0;
    if (seq_.size() != qual_.size()) return -2;
    return (long)seq_.size();
}

}  // namespace vgh
