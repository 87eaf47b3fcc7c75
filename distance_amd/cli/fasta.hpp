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
#include <memory>
#include <sys/mman.h>
#include <sys/stat.h>
#include <string>

namespace cli {

struct FastaRecord {
    std::string id, desc, seq;
    bool has_desc = false;
};

class FastaReader {
public:
    explicit FastaReader(FILE *fh) : fh_(fh) {}
    // parse an in-memory block (whole records, as cut by BlockReader)
    FastaReader(const char *data, size_t len) : fh_(nullptr), mem_(data), mem_len_(len) {}
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
        if (mem_) {  // in-memory block: one memchr per line
            if (mem_pos_ >= mem_len_)
                return false;
            const char *start = mem_ + mem_pos_;
            const char *nl = (const char *)std::memchr(start, '\n', mem_len_ - mem_pos_);
            const size_t n = nl ? (size_t)(nl - start) + 1 : mem_len_ - mem_pos_;
            line_.assign(start, n);
            mem_pos_ += n;
            return true;
        }
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
    const char *mem_ = nullptr;
    size_t mem_len_ = 0, mem_pos_ = 0;
    std::string line_, err_;
    char buf_[1 << 16];
    size_t pos_ = 0, len_ = 0;
};

// A block of whole records: uninitialised storage filled by fread (std::string::resize would zero it first).
struct Block {
    std::unique_ptr<char[]> data;
    const char *view = nullptr;   // the block's bytes when they are a piece of the mapped file (data is empty then)
    size_t len = 0, cap = 0;
    const char *bytes() const { return view ? view : data.get(); }
};

// Cuts a stream into blocks of whole records so that blocks can be parsed in parallel: a block ends
// just before a line that starts with '>' (every such line starts a record for bio's reader).  Each block is read
// straight into its own buffer; only the partial record behind the last cut is copied (to the next block's front).
// A regular file is mapped instead: its blocks are pieces of the mapping (no read into a buffer on the calling thread —
// 1.5 GB of it were a third of the time a 50,000 x 30,000 load took —, the pages are touched by the threads that parse
// them); same cuts.
class BlockReader {
public:
    BlockReader(FILE *fh, size_t target_bytes) : fh_(fh), target_(target_bytes ? target_bytes : 1)
    {
        struct stat st;
        const int fd = fileno(fh);
        const off_t at = ftello(fh);
        if (fd >= 0 && at >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > at) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                map_ = static_cast<const char *>(m);
                map_len_ = (size_t)st.st_size;
                pos_ = (size_t)at;
                (void)madvise(m, map_len_, MADV_SEQUENTIAL);
            }
        }
    }
    ~BlockReader()
    {
        // (the mapping stays: blocks handed out may outlive the reader; the process unmaps it on exit)
    }
    // false at end of input; otherwise `out` holds >= 1 whole record (the rest of the input at EOF)
    bool next(Block &out)
    {
        if (map_) {
            if (pos_ >= map_len_)
                return false;
            size_t end = std::min(map_len_, pos_ + target_), cut = 0;
            for (;;) {
                if (end == map_len_) {
                    cut = map_len_;
                    break;
                }
                // last "\n>" in [pos_, end): everything before the '>' is whole records
                for (size_t p = end; p-- > pos_ + 1;)
                    if (map_[p] == '>' && map_[p - 1] == '\n') {
                        cut = p;
                        break;
                    }
                if (cut > pos_)
                    break;
                end = std::min(map_len_, end + target_);   // one record longer than the target: look further
            }
            Block cur;
            cur.view = map_ + pos_;
            cur.len = cur.cap = cut - pos_;
            pos_ = cut;
            out = std::move(cur);
            return true;
        }
        Block cur;
        cur.cap = carry_.len + target_;
        cur.data.reset(new char[cur.cap]);
        if (carry_.len)
            std::memcpy(cur.data.get(), carry_.data.get(), carry_.len);
        cur.len = carry_.len;
        carry_ = Block();
        for (;;) {
            if (!eof_) {
                if (cur.len == cur.cap) {   // one record longer than the target: grow and keep reading
                    Block bigger;
                    bigger.cap = cur.cap + target_;
                    bigger.data.reset(new char[bigger.cap]);
                    std::memcpy(bigger.data.get(), cur.data.get(), cur.len);
                    bigger.len = cur.len;
                    cur = std::move(bigger);
                }
                const size_t got = std::fread(cur.data.get() + cur.len, 1, cur.cap - cur.len, fh_);
                cur.len += got;
                if (got == 0)
                    eof_ = true;
            }
            if (cur.len == 0)
                return false;
            if (eof_) {
                out = std::move(cur);
                return true;
            }
            // last "\n>" in the buffer: everything before the '>' is whole records
            const char *d = cur.data.get();
            size_t cut = 0;
            for (size_t p = cur.len; p-- > 1;) {
                if (d[p] == '>' && d[p - 1] == '\n') {
                    cut = p;
                    break;
                }
            }
            if (cut > 0) {
                carry_.len = carry_.cap = cur.len - cut;
                carry_.data.reset(new char[carry_.cap ? carry_.cap : 1]);
                std::memcpy(carry_.data.get(), d + cut, carry_.len);
                cur.len = cut;
                out = std::move(cur);
                return true;
            }
        }
    }

private:
    FILE *fh_;
    size_t target_;
    Block carry_;
    bool eof_ = false;
    const char *map_ = nullptr;
    size_t map_len_ = 0, pos_ = 0;
};

}  // namespace cli
