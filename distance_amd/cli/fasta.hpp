// fasta.hpp — FASTA records as the reference reads them: bio 1.6.0 `fasta::Reader::records()`
// (call sites src/fastaio.rs:179-181, 225, 242-243; crate source is not in the container, so the
// tokenisation below follows rust-bio's published reader and is pinned only by the reference's
// single-line test fixtures, src/lib.rs:906-914 / src/fastaio.rs:344-350 — CRLF, blank lines and
// empty ids are parity-unpinned):
//   * a record starts at a line beginning with '>'; anything else there is the error
//     "Expected > at record start.";
//   * id = header up to the first whitespace, description = the rest (trailing whitespace trimmed);
//   * every following line up to the next '>' line or EOF is appended with trailing whitespace
//     trimmed (so "\r\n" works and an empty line adds nothing).
#pragma once
#include <cstdio>
#include <cstring>
#include <string>

namespace cli {

struct FastaRecord {
    std::string id, desc, seq;
    bool has_desc = false;
};

class FastaReader {
public:
    explicit FastaReader(FILE *fh) : fh_(fh) {}
    // 1 = a record was read, 0 = end of input, -1 = format error (message in error())
    int next(FastaRecord &rec)
    {
        rec.id.clear();
        rec.desc.clear();
        rec.seq.clear();
        rec.has_desc = false;
        if (line_.empty()) {
            if (!read_line())
                return 0;
        }
        if (line_[0] != '>') {
            err_ = "Expected > at record start.";
            return -1;
        }
        size_t end = line_.size();
        while (end > 1 && is_space((unsigned char)line_[end - 1]))
            --end;
        size_t p = 1;
        while (p < end && !is_space((unsigned char)line_[p]))
            ++p;
        rec.id.assign(line_, 1, p - 1);
        if (p < end) {
            rec.has_desc = true;
            rec.desc.assign(line_, p + 1, end - p - 1);
        }
        for (;;) {
            if (!read_line()) {
                line_.clear();
                break;
            }
            if (line_[0] == '>')
                break;
            size_t e = line_.size();
            while (e > 0 && is_space((unsigned char)line_[e - 1]))
                --e;
            rec.seq.append(line_, 0, e);
        }
        return 1;
    }
    const std::string &error() const { return err_; }

private:
    static bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
    bool read_line()  // false at EOF with nothing read
    {
        line_.clear();
        for (;;) {
            if (pos_ == len_) {
                len_ = std::fread(buf_, 1, sizeof buf_, fh_);
                pos_ = 0;
                if (len_ == 0)
                    return !line_.empty();
            }
            const char *start = buf_ + pos_;
            const char *nl = (const char *)std::memchr(start, '\n', len_ - pos_);
            if (nl) {
                line_.append(start, nl - start + 1);
                pos_ += (size_t)(nl - start) + 1;
                return true;
            }
            line_.append(start, len_ - pos_);
            pos_ = len_;
        }
    }
    FILE *fh_;
    std::string line_, err_;
    char buf_[1 << 16];
    size_t pos_ = 0, len_ = 0;
};

}  // namespace cli
