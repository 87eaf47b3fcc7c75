// distance — drop-in CLI for benjamincjackson/distance on MI355X.
//
// Keeps the reference's surface (clap definition src/lib.rs:68-131, rules of set_up()
// src/lib.rs:162-267): -i/--input (0..2 files) or positional files, -s/--stream (file or "-"),
// -m/--measure {n,n_high,raw,jc69,k80,tn93} (default raw), -o/--output, -t/--threads,
// -b/--batchsize, -l/--licenses, -h, -V; TSV output identical to gather_write()
// (src/lib.rs:612-644).  The pair generators and worker pools (src/lib.rs:269-474, 502-596) are
// replaced by libdistance_hip.so through its C ABI; -t sizes the host formatting pool and -b is
// accepted — neither changes the output, as in the reference (src/lib.rs:919-1154).
// Extra flags: --gpus N (default 1) / --devices LIST, --slab-pairs P (result slab size).
//
// Exactness: the GPU returns integer site tallies; f64 finalisation is dst_finalize() on the host
// (reference operation order, glibc log/sqrt), so the printed digits do not depend on the device.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <cstdio>
#include <sys/stat.h>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../include/distance_hip.h"
#include "fasta.hpp"
#include "format.hpp"

namespace {

const char *kVersion = "0.3.1";  // Cargo.toml:3 of the reference this CLI mirrors

// ---------------------------------------------------------------- errors ----------------------
// main() of the reference returns Result<(), DistanceError>: Rust prints `Error: {:?}` and exits 1.
// Fatal errors can surface on a reader / worker thread while GPU threads are still running: flush
// what was written (the reference's BufWriter content is flushed on its way out too) and leave
// without running static destructors under those threads.
[[noreturn]] void leave(int code)
{
    std::fflush(nullptr);
    _exit(code);
}

[[noreturn]] void die_message(const std::string &m)
{
    std::fprintf(stderr, "Error: Message(\"%s\")\n", m.c_str());
    leave(1);
}

[[noreturn]] void die_io(const std::string &what, int err)
{
    std::fprintf(stderr, "Error: IOError(Os { code: %d, kind: %s, message: \"%s\" }) [%s]\n", err,
                 err == ENOENT ? "NotFound" : err == EACCES ? "PermissionDenied" : "Other", std::strerror(err),
                 what.c_str());
    leave(1);
}

[[noreturn]] void die_usage(const std::string &m)
{
    std::fprintf(stderr, "error: %s\n\nFor more information, try '--help'.\n", m.c_str());
    std::exit(2);  // clap's usage-error exit code
}

void print_help()
{
    std::puts(
        "Calculate genetic distances within/between fasta-format alignments of DNA sequences\n\n"
        "Usage: All sequences across all input files must be the same length.\n\n"
        "       distance alignment.fasta\n"
        "       cat alignment.fasta | distance\n"
        "       distance alignment.fasta -o distances.tsv\n"
        "       distance -t 8 -m jc69 alignment.fasta -o jc69.tsv\n"
        "       distance alignment1.fasta alignment2.fasta > distances2.tsv\n"
        "       distance -i smallAlignment.fasta -s bigAlignment.fasta -o distances3.tsv\n"
        "       cat bigAlignment.fasta | distance smallAlignment.fasta -s - > distances3.tsv\n"
        "       \n\n"
        "Options:\n"
        "  -i, --input [<input>...]     One or two input alignment files in fasta format. Loaded into memory. "
        "This flag can be omitted and the files passed as positional arguments\n"
        "  -s, --stream <stream>        One input alignment file in fasta format. Streamed from disk (or stdin "
        "using \"-s -\"). Requires exactly one file also be loaded\n"
        "  -m, --measure <measure>      Which distance measure to use [default: raw] [possible values: n, n_high, "
        "raw, jc69, k80, tn93]\n"
        "  -o, --output <output>        Output file in tab-separated-value format. Omit this option to print to "
        "stdout\n"
        "  -t, --threads <threads>      How many threads to spin up for pairwise comparisons. Omitting this option "
        "spins up the number of available CPUs\n"
        "  -b, --batchsize <batchsize>  Try setting this >(>) 1 to tune the workload per thread [default: 1]\n"
        "  -l, --licenses               Print licence information and exit\n"
        "      --gpus <n>               MI355X GPUs to use (default 1)\n"
        "      --devices <list>         Explicit device ordinals, e.g. 0,1,2,3 (overrides --gpus)\n"
        "  -h, --help                   Print help\n"
        "  -V, --version                Print version");
}

// ---------------------------------------------------------------- arguments -------------------
struct Args {
    std::vector<std::string> flag_inputs, pos_inputs;
    bool has_stream = false;
    std::string stream, measure = "raw", output;
    bool has_output = false, has_threads = false, licenses = false;
    size_t threads = 0, batchsize = 1;
    int gpus = 1;
    std::vector<int> devices;  // explicit device ordinals (--devices 0,1,..); empty: 0..gpus-1
    size_t slab_pairs = (size_t)1 << 22;  // result slab: 4 Mi pairs pipelines GPU, formatters and writer well
    std::string selftest;
};

size_t parse_usize(const std::string &v, const char *flag)
{
    if (v.empty() || v.find_first_not_of("0123456789") != std::string::npos)
        die_usage("invalid value '" + v + "' for '" + flag + "': invalid digit found in string");
    errno = 0;
    unsigned long long x = std::strtoull(v.c_str(), nullptr, 10);
    if (errno)
        die_usage("invalid value '" + v + "' for '" + flag + "': number too large to fit in target type");
    return (size_t)x;
}

Args parse_args(int argc, char **argv)
{
    Args a;
    auto value_of = [&](int &k, const std::string &arg, const char *flag) -> std::string {
        const size_t eq = arg.find('=');
        if (arg.rfind("--", 0) == 0 && eq != std::string::npos)
            return arg.substr(eq + 1);
        if (arg.rfind("--", 0) != 0 && arg.size() > 2)
            return arg.substr(2);  // -mraw
        if (k + 1 >= argc)
            die_usage(std::string("a value is required for '") + flag + "' but none was supplied");
        return argv[++k];
    };
    auto is = [](const std::string &arg, const char *s, const char *l) {
        if (arg == l || arg.rfind(std::string(l) + "=", 0) == 0)
            return true;
        return arg.rfind(s, 0) == 0 && arg.rfind("--", 0) != 0;
    };
    bool only_pos = false;
    for (int k = 1; k < argc; ++k) {
        const std::string arg = argv[k];
        if (only_pos || arg == "-" || arg.empty() || arg[0] != '-') {
            a.pos_inputs.push_back(arg);
            continue;
        }
        if (arg == "--") {
            only_pos = true;
        } else if (arg == "-h" || arg == "--help") {
            print_help();
            std::exit(0);
        } else if (arg == "-V" || arg == "--version") {
            std::printf("distance %s\n", kVersion);
            std::exit(0);
        } else if (arg == "-l" || arg == "--licenses") {
            a.licenses = true;
        } else if (is(arg, "-i", "--input")) {
            // num_args(0..=2): takes following non-flag words, at most two
            const size_t eq = arg.find('=');
            if (arg.rfind("--", 0) == 0 && eq != std::string::npos)
                a.flag_inputs.push_back(arg.substr(eq + 1));
            else if (arg.rfind("--", 0) != 0 && arg.size() > 2)
                a.flag_inputs.push_back(arg.substr(2));
            while (a.flag_inputs.size() < 2 && k + 1 < argc && (argv[k + 1][0] != '-' || !std::strcmp(argv[k + 1], "-")))
                a.flag_inputs.push_back(argv[++k]);
        } else if (is(arg, "-s", "--stream")) {
            a.stream = value_of(k, arg, "--stream <stream>");
            a.has_stream = true;
        } else if (is(arg, "-m", "--measure")) {
            a.measure = value_of(k, arg, "--measure <measure>");
        } else if (is(arg, "-o", "--output")) {
            a.output = value_of(k, arg, "--output <output>");
            a.has_output = true;
        } else if (is(arg, "-t", "--threads")) {
            a.threads = parse_usize(value_of(k, arg, "--threads <threads>"), "--threads <threads>");
            a.has_threads = true;
        } else if (is(arg, "-b", "--batchsize")) {
            a.batchsize = parse_usize(value_of(k, arg, "--batchsize <batchsize>"), "--batchsize <batchsize>");
        } else if (arg == "--gpus" || arg.rfind("--gpus=", 0) == 0) {
            a.gpus = (int)parse_usize(value_of(k, arg, "--gpus <n>"), "--gpus <n>");
        } else if (arg == "--devices" || arg.rfind("--devices=", 0) == 0) {
            std::string v = value_of(k, arg, "--devices <list>");
            for (size_t p = 0; p <= v.size();) {
                const size_t q = std::min(v.find(',', p), v.size());
                a.devices.push_back((int)parse_usize(v.substr(p, q - p), "--devices <list>"));
                p = q + 1;
            }
        } else if (arg == "--slab-pairs" || arg.rfind("--slab-pairs=", 0) == 0) {
            a.slab_pairs = std::max<size_t>(1, parse_usize(value_of(k, arg, "--slab-pairs <p>"), "--slab-pairs <p>"));
        } else if (arg == "--host-selftest") {
            a.selftest = value_of(k, arg, "--host-selftest <what>");
        } else {
            die_usage("unexpected argument '" + arg + "' found");
        }
    }
    if (a.pos_inputs.size() > 2)
        die_usage("unexpected argument '" + a.pos_inputs[2] + "' found");
    if (dst_measure_from_name(a.measure.c_str()) < 0)
        die_usage("invalid value '" + a.measure + "' for '--measure <measure>'\n  [possible values: n, n_high, raw, "
                  "jc69, k80, tn93]");
    return a;
}

const char *kLicences =
    "\nCopyright 2022, Ben Jackson (distance, GNU LIBRARY GENERAL PUBLIC LICENSE, Version 2): this program is an\n"
    "independent MI355X re-implementation of its command-line surface and output format.\n"
    "FASTA tokenisation follows Rust-Bio (MIT licence, Copyright (c) 2016 Johannes Koester, the Rust-Bio team,\n"
    "Google Inc.).  Nucleotide coding scheme: Emmanuel Paradis, as used in ape.\n";

// ---------------------------------------------------------------- loading ---------------------
// src/encoding.rs:4-41
void encoding_array(uint8_t a[256])
{
    std::memset(a, 0, 256);
    const char *letters = "AGCTRMWSKYVHDBN";
    const uint8_t codes[] = {136, 72, 40, 24, 192, 160, 144, 96, 80, 48, 224, 176, 208, 112, 240};
    for (int k = 0; letters[k]; ++k) {
        a[(unsigned char)letters[k]] = codes[k];
        a[(unsigned char)(letters[k] - 'A' + 'a')] = codes[k];
    }
    a[(unsigned char)'-'] = 244;
    a[(unsigned char)'?'] = 242;
}

// std::vector that leaves new bytes uninitialised (resize() before a fill would touch every page twice)
template <class T>
struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <class U> NoInit(const NoInit<U> &) {}
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
using Bytes = std::vector<uint8_t, NoInit<uint8_t>>;

struct Alignment {
    std::vector<std::string> ids;
    Bytes codes;                     // n x width, row-major
    std::vector<uint32_t> counts;    // n x 4 {A,T,G,C}: only filled for streamed tn93 batches
    size_t n = 0, width = 0;
};

std::string err_invalid_nuc(const std::string &id, unsigned char c)  // src/fastaio.rs:89-91
{
    std::string s = "Invalid nucleotide character in record '" + id + "': '";
    s.push_back((char)c);
    return s + "'";
}

std::string err_lengths(size_t w1, size_t w2)  // src/fastaio.rs:93-95
{
    return "Different length sequences in alignment(s): " + std::to_string(w1) + " vs " + std::to_string(w2);
}

FILE *open_input(const std::string &path)
{
    FILE *fh = std::fopen(path.c_str(), "rb");
    if (!fh)
        die_io(path, errno);
    return fh;
}

// One block of whole records -> encoded records.  Errors are returned, not raised, so that the
// consumer reports the FIRST one in file order, like the reference's sequential reader.
struct ParsedBlock {
    std::unique_ptr<Alignment> al;
    std::string error;  // empty: ok
    bool stop = false;  // an empty record ended bio's Records iterator: nothing after it is read
};

ParsedBlock parse_block(const cli::Block &block, const uint8_t *table, bool count_raw_upper, bool fixed_width,
                        size_t width, bool width_first)
{
    // In place on the block's bytes, with fasta.hpp's tokenisation (FastaReader is the readable statement of it and
    // the sequential reader the host tests compare this with): a record starts at a line beginning with '>';
    // id = header up to the first whitespace, the rest (trailing whitespace trimmed) is the description; every
    // following line up to the next '>' line is sequence, trailing whitespace trimmed.  No per-line copies: the
    // lines are encoded straight from the block into the codes.
    ParsedBlock out;
    out.al = std::make_unique<Alignment>();
    Alignment &al = *out.al;
    al.width = width;
    al.codes.reserve(block.len);   // an upper bound: a code per sequence byte
    const char *data = block.bytes();
    const size_t len = block.len;
    auto is_space = [](unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); };
    bool first = !fixed_width;
    std::vector<std::pair<const char *, size_t>> parts;
    std::string id;
    size_t pos = 0;
    while (pos < len) {
        // header line
        const char *h = data + pos;
        const char *nl = (const char *)std::memchr(h, '\n', len - pos);
        size_t hlen = nl ? (size_t)(nl - h) + 1 : len - pos;
        if (h[0] != '>') {
            out.error = "Expected > at record start.";
            return out;
        }
        pos += hlen;
        size_t end = hlen;
        while (end > 1 && is_space((unsigned char)h[end - 1]))
            --end;
        size_t p = 1;
        while (p < end && !is_space((unsigned char)h[p]))
            ++p;
        id.assign(h + 1, p - 1);
        const bool has_desc = p < end;
        // sequence lines
        parts.clear();
        size_t total = 0;
        while (pos < len && data[pos] != '>') {
            const char *l = data + pos;
            const char *e = (const char *)std::memchr(l, '\n', len - pos);
            const size_t llen = e ? (size_t)(e - l) + 1 : len - pos;
            pos += llen;
            size_t keep = llen;
            while (keep > 0 && is_space((unsigned char)l[keep - 1]))
                --keep;
            if (keep) {
                parts.emplace_back(l, keep);
                total += keep;
            }
        }
        if (id.empty() && !has_desc && total == 0) {
            out.stop = true;  // bio's Records iterator stops at an empty record
            break;
        }
        // stream_fasta() compares widths BEFORE encoding (src/fastaio.rs:246-254); load_fasta() encodes first
        // (src/fastaio.rs:182) and compares afterwards (:186-190), so a loaded record that is both the wrong
        // length and holds an invalid character reports the character.
        if (width_first && total != al.width) {
            out.error = err_lengths(total, al.width);  // src/fastaio.rs:93-95, 246-248
            return out;
        }
        // encode() / encode_count_bases(): src/fastaio.rs:101-145
        const size_t at = al.codes.size();
        al.codes.resize(at + total);
        uint8_t *dst = al.codes.data() + at;
        uint32_t counting[256];
        if (count_raw_upper)
            std::memset(counting, 0, sizeof counting);
        for (const auto &part : parts) {
            const unsigned char *src = (const unsigned char *)part.first;
            const size_t n = part.second;
            uint8_t bad = 0xFF;   // AND of the codes: 0 as soon as one byte has none (every code has bit 7, 6, 5 or 4)
            for (size_t i = 0; i < n; ++i) {
                const uint8_t code = table[src[i]];
                dst[i] = code;
                bad = code ? bad : 0;
            }
            if (!bad) {
                for (size_t i = 0; i < n; ++i)
                    if (table[src[i]] == 0) {
                        out.error = err_invalid_nuc(id, src[i]);
                        return out;
                    }
            }
            if (count_raw_upper)
                for (size_t i = 0; i < n; ++i)
                    counting[src[i]] += 1;
            dst += n;
        }
        if (count_raw_upper) {
            al.counts.push_back(counting['A']);
            al.counts.push_back(counting['T']);
            al.counts.push_back(counting['G']);
            al.counts.push_back(counting['C']);
        }
        if (!width_first) {
            if (first) {
                al.width = total;
                first = false;
            } else if (total != al.width) {
                out.error = err_lengths(total, al.width);  // src/fastaio.rs:93-95, 186-190
                return out;
            }
        }
        al.ids.push_back(id);
        al.n += 1;
    }
    return out;
}

// Reads `fh` block by block and parses up to `lookahead` blocks concurrently; `consume` gets the
// parsed blocks strictly in file order (and is where errors surface).
template <class Consume>
void parse_stream(FILE *fh, size_t block_bytes, size_t lookahead, const uint8_t *table, bool count_raw_upper,
                  bool fixed_width, size_t width, Consume consume)
{
    cli::BlockReader blocks(fh, block_bytes);
    std::deque<std::future<ParsedBlock>> inflight;
    cli::Block block;
    bool more = true, width_known = fixed_width;
    size_t w = width;
    while (more || !inflight.empty()) {
        // the first block runs alone when the width is not known yet (it fixes it for the others)
        while (more && inflight.size() < (width_known ? std::max<size_t>(lookahead, 1) : 1)) {
            if (!blocks.next(block)) {
                more = false;
                break;
            }
            inflight.push_back(std::async(std::launch::async,
                                          [b = std::make_shared<cli::Block>(std::move(block)), table, count_raw_upper,
                                           width_known, w, fixed_width]() {
                                              return parse_block(*b, table, count_raw_upper, width_known, w, fixed_width);
                                          }));
            block = cli::Block();
        }
        if (inflight.empty())
            break;
        ParsedBlock pb = inflight.front().get();
        inflight.pop_front();
        if (!pb.error.empty())
            die_message(pb.error);
        if (!width_known && pb.al->n) {
            width_known = true;
            w = pb.al->width;
        }
        const bool stop = pb.stop;
        consume(std::move(pb.al));
        if (stop) {
            for (auto &f : inflight)
                f.wait();
            return;
        }
    }
}

// load_fasta(): src/fastaio.rs:174-199 (blocks parsed in parallel, assembled in file order)
Alignment load_fasta(FILE *fh, const uint8_t *table, size_t threads)
{
    Alignment al;
    bool first = true;
    // DISTANCE_PARSE_BLOCK_BYTES: test hook (tiny blocks put the records of a small file into different parse blocks)
    const char *bb = std::getenv("DISTANCE_PARSE_BLOCK_BYTES");
    const size_t block_bytes = bb && std::atol(bb) > 0 ? (size_t)std::atol(bb) : (size_t)32 << 20;
    // the codes are fewer than the file's bytes: one allocation, no growth copies of a multi-GB vector
    struct stat st;
    if (fstat(fileno(fh), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0)
        al.codes.reserve((size_t)st.st_size);
    // a part's codes are copied into place off this thread (it reads the next block meanwhile: the copy of 1.5 GB was a
    // third of the time this loop took) — as long as the reserved buffer holds them: a growing buffer moves
    std::deque<std::future<void>> copies;
    parse_stream(fh, block_bytes, threads, table, false, false, 0, [&](std::unique_ptr<Alignment> part) {
        if (part->n == 0)
            return;
        if (first) {
            al.width = part->width;
            first = false;
        }
        const size_t at = al.codes.size(), bytes = part->codes.size();
        const bool in_place = at + bytes <= al.codes.capacity();
        if (!in_place) {   // the buffer is about to move: nothing may still be writing into it
            for (auto &c : copies)
                c.get();
            copies.clear();
        }
        al.codes.resize(at + bytes);
        for (auto &id : part->ids)
            al.ids.push_back(std::move(id));
        al.n += part->n;
        if (in_place) {
            while (copies.size() >= 4) {
                copies.front().get();
                copies.pop_front();
            }
            uint8_t *dst = al.codes.data() + at;
            copies.push_back(std::async(std::launch::async, [dst, bytes, p = std::shared_ptr<Alignment>(std::move(part))] {
                std::memcpy(dst, p->codes.data(), bytes);
            }));
        } else {
            std::memcpy(al.codes.data() + at, part->codes.data(), bytes);
        }
    });
    for (auto &c : copies)
        c.get();
    if (al.n == 0)
        die_message("Empty FASTA file");  // src/fastaio.rs:97-99
    return al;
}

// ---------------------------------------------------------------- ordered output --------------
struct Writer {
    FILE *fh = stdout;
    void write(const char *p, size_t n)
    {
        if (!n)
            return;
        if (std::fwrite(p, 1, n, fh) != n) {
            if (errno == EPIPE)
                _exit(0);  // handle_broken_pipe(): src/lib.rs:598-608
            die_io("write", errno);
        }
    }
    void flush()
    {
        if (std::fflush(fh) != 0) {
            if (errno == EPIPE)
                _exit(0);
            die_io("flush", errno);
        }
    }
};

struct Ctx {
    dst_ctx *h = nullptr;
    void check(int rc, const char *what) const
    {
        if (rc != DST_OK) {
            std::fprintf(stderr, "Error: Gpu(\"%s: %s\")\n", what, dst_last_error(h));
            leave(1);
        }
    }
};

// One slab of results: rows [rb, re) of the row set against the column set (square: j > i).
// Text buffers come from a pool: a fresh multi-megabyte allocation is mmap'ed and page-faulted in on every slab
// (the kernel zeroes each page), which cost a fifth of the formatting time at 1e9 lines.
class TextPool {
public:
    ~TextPool()
    {
        for (auto &b : free_)
            delete[] b.first;
    }
    char *acquire(size_t need, size_t *cap)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            size_t best = free_.size();
            for (size_t k = 0; k < free_.size(); ++k)
                if (free_[k].second >= need && (best == free_.size() || free_[k].second < free_[best].second))
                    best = k;
            if (best != free_.size()) {
                auto b = free_[best];
                free_.erase(free_.begin() + (long)best);
                *cap = b.second;
                return b.first;
            }
        }
        *cap = need;
        return new char[need];
    }
    void release(char *p, size_t cap)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (free_.size() >= 64) {  // bound what is kept
            delete[] p;
            return;
        }
        free_.emplace_back(p, cap);
    }

private:
    std::mutex mu_;
    std::vector<std::pair<char *, size_t>> free_;
};
TextPool g_text_pool;

// un-initialised growable char buffer (std::string::resize would zero-fill every byte first)
struct TextBuf {
    struct Ptr {  // minimal owner so that the call sites keep reading `out.p.get()`
        char *raw = nullptr;
        char *get() const { return raw; }
    } p;
    size_t len = 0, cap = 0;
    TextBuf() = default;
    TextBuf(const TextBuf &) = delete;
    TextBuf &operator=(const TextBuf &) = delete;
    TextBuf(TextBuf &&o) noexcept : p(o.p), len(o.len), cap(o.cap)
    {
        o.p.raw = nullptr;
        o.len = o.cap = 0;
    }
    TextBuf &operator=(TextBuf &&o) noexcept
    {
        if (this != &o) {
            drop();
            p = o.p;
            len = o.len;
            cap = o.cap;
            o.p.raw = nullptr;
            o.len = o.cap = 0;
        }
        return *this;
    }
    ~TextBuf() { drop(); }
    void drop()
    {
        if (p.raw)
            g_text_pool.release(p.raw, cap);
        p.raw = nullptr;
        len = cap = 0;
    }
    void ensure(size_t need)
    {
        if (len + need <= cap)
            return;
        size_t ncap = 0;
        char *q = g_text_pool.acquire(std::max(cap + cap / 2, len + need + ((size_t)1 << 16)), &ncap);
        if (len)
            std::memcpy(q, p.raw, len);
        if (p.raw)
            g_text_pool.release(p.raw, cap);
        p.raw = q;
        cap = ncap;
    }
};

// page-locked result buffers, recycled between slabs (pinning pages costs more than the copy)
// Page-locking a 200 MB buffer costs ~40 ms and unlocking it ~25: the pool hands out buffers of ONE size (the caller asks
// for its largest slab: slabs of whole rows differ by a row's worth, and a buffer a few bytes short used to mean a new
// one — 0.3 s of a 2 s run at 50,000 x 30,000), a helper thread brings the first ones up while the sets are uploaded
// (prewarm), and nothing is unlocked on the way out: the process ends right after the last slab.
class PinnedPool {
public:
    ~PinnedPool()
    {
        if (warm_.joinable())
            warm_.join();
    }
    void prewarm(size_t count, size_t bytes)
    {
        pending_ = count;
        warm_ = std::thread([this, count, bytes] {
            for (size_t k = 0; k < count; ++k) {
                void *p = nullptr;
                const bool ok = dst_host_alloc(bytes, &p) == DST_OK;
                std::lock_guard<std::mutex> lk(mu_);
                if (ok)
                    free_.emplace_back(p, bytes);
                --pending_;
                cv_.notify_all();
            }
        });
    }
    uint32_t *acquire(size_t bytes, size_t *cap)
    {
        {
            std::unique_lock<std::mutex> lk(mu_);
            for (;;) {
                for (size_t k = 0; k < free_.size(); ++k)
                    if (free_[k].second >= bytes) {
                        auto b = free_[k];
                        free_.erase(free_.begin() + (long)k);
                        *cap = b.second;
                        return static_cast<uint32_t *>(b.first);
                    }
                if (pending_ == 0)
                    break;
                cv_.wait(lk);   // one is on its way
            }
        }
        void *p = nullptr;
        if (dst_host_alloc(bytes, &p) != DST_OK)
            return nullptr;
        *cap = bytes;
        return static_cast<uint32_t *>(p);
    }
    void release(uint32_t *p, size_t cap)
    {
        std::lock_guard<std::mutex> lk(mu_);
        free_.emplace_back(p, cap);
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::pair<void *, size_t>> free_;
    size_t pending_ = 0;
    std::thread warm_;
};

struct Slab {
    uint64_t rb = 0, re = 0;
    uint32_t *tallies = nullptr;  // page-locked (PinnedPool): the copy back runs at link speed
    size_t tallies_cap = 0;
    std::vector<TextBuf> text;            // formatted parts, in canonical order
    char *gtext = nullptr;                // the whole slab's text, formatted on the GPU (page-locked, PinnedPool)
    size_t gtext_len = 0, gtext_cap = 0;
};

struct Job {
    int measure = 0;
    bool square = true;
    bool swap_ids = false;            // stream mode: rows are the streamed batch, id1 is the loaded id
    const Alignment *rows = nullptr;  // row set (ids + per-record counts)
    const Alignment *cols = nullptr;
    const uint32_t *row_counts = nullptr, *col_counts = nullptr;  // tn93 {A,T,G,C}
    size_t fmt_threads = 1;
    bool gpu_text = false;            // the GPU writes the TSV lines itself (dst_text_*); the host only writes them out
};

// tallies -> TSV text, in canonical order, split over the formatting pool (-t)
void format_slab(const Job &job, Slab &slab)
{
    const int w = dst_tally_width(job.measure);
    const uint64_t ncols = job.cols->n;
    const uint64_t rows = slab.re - slab.rb;
    const size_t T = std::max<size_t>(1, std::min<size_t>(job.fmt_threads, rows));
    slab.text.clear();
    slab.text.resize(T);
    // split rows so that each part has about the same number of pairs
    std::vector<uint64_t> cut(T + 1, slab.re);
    cut[0] = slab.rb;
    {
        auto pairs_before = [&](uint64_t r) -> uint64_t {
            if (!job.square)
                return (r - slab.rb) * ncols;
            return dst_square_row_start(ncols, r) - dst_square_row_start(ncols, slab.rb);
        };
        const uint64_t total = pairs_before(slab.re);
        uint64_t r = slab.rb;
        for (size_t k = 1; k < T; ++k) {
            const uint64_t target = total * k / T;
            while (r < slab.re && pairs_before(r) < target)
                ++r;
            cut[k] = r;
        }
    }
    // hot loop of the host side: one TSV line per pair, written with raw pointer bumps into a
    // buffer that is grown geometrically (std::string::append per field costs ~4x more)
    const bool is_int = job.measure == DST_N || job.measure == DST_N_HIGH;
    auto pairs_between = [&](uint64_t r0, uint64_t r1) -> uint64_t {
        if (!job.square)
            return (r1 - r0) * ncols;
        return dst_square_row_start(ncols, r1) - dst_square_row_start(ncols, r0);
    };
    size_t id_max = 0;
    for (const auto &id : job.cols->ids)
        id_max = std::max(id_max, id.size());
    // raw / jc69 / k80: the distance is a function of the pair's tallies alone, and the same tallies recur all the
    // time (a few hundred distinct (n, d) in a SARS-CoV-2-like alignment): a small direct-mapped memo of
    // tallies -> printed number takes most pairs past dst_finalize and the exact decimal conversion.  A miss
    // computes it the ordinary way, so the text is the same either way.
    const bool memo_ok = job.measure == DST_RAW || job.measure == DST_JC69 || job.measure == DST_K80;
    struct Memo {
        uint64_t key_lo;
        uint32_t key_hi;
        uint8_t len;   // 0: empty
        char text[27];
    };
    constexpr size_t kMemoSize = 1u << 14;
    auto work = [&](size_t k) {
        TextBuf &out = slab.text[k];
        std::vector<Memo> memo(memo_ok ? kMemoSize : 0);
        for (auto &mslot : memo)
            mslot.len = 0;
        // first guess: 16 characters of number per line; grows (rarely) if a line needs more
        out.ensure((size_t)pairs_between(cut[k], cut[k + 1]) * (2 * id_max + 20) + 64);
        for (uint64_t i = cut[k]; i < cut[k + 1]; ++i) {
            const uint64_t j0 = job.square ? i + 1 : 0;
            uint64_t p = job.square ? dst_square_row_start(ncols, i) - dst_square_row_start(ncols, slab.rb)
                                    : (i - slab.rb) * ncols;
            const std::string &row_id = job.rows->ids[i];
            const uint32_t *rc = job.row_counts ? job.row_counts + 4 * i : nullptr;
            for (uint64_t j = j0; j < ncols; ++j, ++p) {
                const uint32_t *tl = &slab.tallies[p * w];
                const std::string &id1 = job.swap_ids ? job.cols->ids[j] : row_id;
                const std::string &id2 = job.swap_ids ? row_id : job.cols->ids[j];
                out.ensure(id1.size() + id2.size() + 3 + cli::kFixed12Max);
                char *o = out.p.get() + out.len;
                std::memcpy(o, id1.data(), id1.size());
                o += id1.size();
                *o++ = '\t';
                std::memcpy(o, id2.data(), id2.size());
                o += id2.size();
                *o++ = '\t';
                Memo *slot = nullptr;
                uint64_t key_lo = 0;
                uint32_t key_hi = 0;
                if (memo_ok) {
                    key_lo = (uint64_t)tl[0] | (uint64_t)tl[1] << 32;
                    key_hi = w > 2 ? tl[2] : 0u;
                    slot = &memo[(size_t)((key_lo * 0x9E3779B97F4A7C15ull ^ (uint64_t)key_hi * 0xC2B2AE3D27D4EB4Full) >> 50)];
                    if (slot->len && slot->key_lo == key_lo && slot->key_hi == key_hi) {
                        std::memcpy(o, slot->text, sizeof slot->text);  // fixed-size copy; only `len` bytes count
                        o += slot->len;
                        *o++ = '\n';
                        out.len = (size_t)(o - out.p.get());
                        continue;
                    }
                }
                double f = 0;
                int64_t iv = 0;
                const uint32_t *cc = job.col_counts ? job.col_counts + 4 * j : nullptr;
                // record_1 = loaded / file-0 record, record_2 = the other (src/lib.rs:325, 432-434)
                if (job.swap_ids)
                    dst_finalize(job.measure, tl, cc, rc, &f, &iv);
                else
                    dst_finalize(job.measure, tl, rc, cc, &f, &iv);
                const int len = is_int ? cli::fmt_i64(iv, o) : cli::fmt_fixed12(f, o);
                if (slot && (size_t)len <= sizeof slot->text) {
                    slot->key_lo = key_lo;
                    slot->key_hi = key_hi;
                    slot->len = (uint8_t)len;
                    std::memcpy(slot->text, o, (size_t)len);
                }
                o += len;
                *o++ = '\n';
                out.len = (size_t)(o - out.p.get());
            }
        }
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (size_t k = 1; k < T; ++k)
            th.emplace_back(work, k);
        work(0);
        for (auto &t : th)
            t.join();
    }
}

// rows [0, n_rows) cut into slabs of <= max_pairs pairs (at least one row each)
std::vector<std::pair<uint64_t, uint64_t>> make_slabs(bool square, uint64_t n_rows, uint64_t n_cols, uint64_t max_pairs)
{
    std::vector<std::pair<uint64_t, uint64_t>> out;
    uint64_t rb = 0;
    const uint64_t last = square ? (n_rows ? n_rows - 1 : 0) : n_rows;  // the last square row has no pairs
    while (rb < last) {
        uint64_t re = rb, pairs = 0;
        while (re < last) {
            const uint64_t row_pairs = square ? n_cols - re - 1 : n_cols;
            if (re > rb && pairs + row_pairs > max_pairs)
                break;
            pairs += row_pairs;
            ++re;
        }
        out.emplace_back(rb, re);
        rb = re;
    }
    return out;
}

// Compute every slab on the GPUs (slab k on GPU k mod G), format on the host pool, write in order.
// Three stages, like the reference's generator -> workers -> gather_write (src/lib.rs:377-458):
//   GPU threads (one per context): slab k's tallies into page-locked memory, slab k on context k mod G;
//   formatter threads: tallies -> exact TSV text (each fans out over its share of the -t pool);
//   the calling thread: writes slabs strictly in canonical order (gather_write's idx re-ordering).
void run_slabs(std::vector<Ctx> &gpus, const Job &job, int row_slot, int col_slot, uint64_t max_pairs, Writer &wr)
{
    const auto slabs = make_slabs(job.square, job.rows->n, job.cols->n, max_pairs);
    const int w = dst_tally_width(job.measure);
    std::mutex mu;
    std::condition_variable cv;
    size_t next_to_write = 0;
    std::atomic<size_t> next_slab{0};
    const size_t n_formatters = 2;
    // bound on slabs in flight (page-locked memory; pinning a 240 MB text buffer costs ~40 ms): with the GPU writing
    // the text a slab is either being copied back or being written out
    const size_t window = job.gpu_text ? gpus.size() * 2 + 1 : gpus.size() * 2 + n_formatters + 1;
    PinnedPool pool;
    std::deque<std::pair<size_t, std::unique_ptr<Slab>>> computed;  // GPU done, waiting for a formatter
    size_t gpu_threads_left = gpus.size();
    std::vector<std::unique_ptr<Slab>> ready(slabs.size());
    Job fjob = job;
    fjob.fmt_threads = std::max<size_t>(1, job.fmt_threads / n_formatters);
    size_t row_id_max = 0, col_id_max = 0;
    for (const auto &id : job.rows->ids)
        row_id_max = std::max(row_id_max, id.size());
    for (const auto &id : job.cols->ids)
        col_id_max = std::max(col_id_max, id.size());
    // every text buffer has the size of the largest slab (ids + number + separators per line)
    uint64_t slab_pairs_max = 0;
    for (const auto &sl : slabs)
        slab_pairs_max = std::max<uint64_t>(slab_pairs_max, job.square ? dst_square_row_start(job.cols->n, sl.second) -
                                                                              dst_square_row_start(job.cols->n, sl.first)
                                                                        : (sl.second - sl.first) * job.cols->n);
    const size_t text_bytes = (size_t)slab_pairs_max * (row_id_max + col_id_max + 34) + 64;
    if (job.gpu_text && !slabs.empty())
        pool.prewarm(std::min(window, slabs.size()), text_bytes);

    auto gpu_worker = [&](size_t g) {
        for (;;) {
            const size_t k = next_slab.fetch_add(1);
            if (k >= slabs.size())
                break;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return k < next_to_write + window; });
            }
            auto s = std::make_unique<Slab>();
            s->rb = slabs[k].first;
            s->re = slabs[k].second;
            const uint64_t pairs = job.square ? dst_square_row_start(job.cols->n, s->re) - dst_square_row_start(job.cols->n, s->rb)
                                              : (s->re - s->rb) * job.cols->n;
            if (job.gpu_text) {
                // ids + number + separators per line; a slab the device formatter declines (a value without a short
                // text, a slab beyond its limits) is formatted on the host like before
                size_t cap = 0, len = 0;
                (void)pairs;
                char *buf = reinterpret_cast<char *>(pool.acquire(text_bytes, &cap));
                if (!buf)
                    gpus[g].check(DST_ERR_NOMEM, "pinned host buffer");
                const int trc = job.square ? dst_text_square(gpus[g].h, job.measure, s->rb, s->re, buf, cap, &len)
                                           : dst_text_rect(gpus[g].h, job.measure, row_slot, col_slot, s->rb, s->re, 0, buf,
                                                           cap, &len);
                if (trc == DST_OK) {
                    s->gtext = buf;
                    s->gtext_len = len;
                    s->gtext_cap = cap;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        ready[k] = std::move(s);
                    }
                    cv.notify_all();
                    continue;
                }
                pool.release(reinterpret_cast<uint32_t *>(buf), cap);
                if (trc != DST_ERR_STATE && trc != DST_ERR_ARG && trc != DST_ERR_CAPACITY)
                    gpus[g].check(trc, "text");
            }
            const size_t n_tallies = (size_t)pairs * w;
            s->tallies = pool.acquire(std::max<size_t>(n_tallies, 1) * 4, &s->tallies_cap);
            if (!s->tallies)
                gpus[g].check(DST_ERR_NOMEM, "pinned host buffer");
            const int rc = job.square ? dst_run_square_host(gpus[g].h, job.measure, s->rb, s->re, DST_OUT_TALLY,
                                                            s->tallies, n_tallies * 4)
                                      : dst_run_rect_host(gpus[g].h, job.measure, row_slot, col_slot, s->rb, s->re,
                                                          DST_OUT_TALLY, s->tallies, n_tallies * 4);
            gpus[g].check(rc, "run");
            {
                std::lock_guard<std::mutex> lk(mu);
                computed.emplace_back(k, std::move(s));
            }
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        --gpu_threads_left;
        cv.notify_all();
    };
    auto formatter = [&]() {
        for (;;) {
            std::pair<size_t, std::unique_ptr<Slab>> item;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !computed.empty() || gpu_threads_left == 0; });
                if (computed.empty())
                    return;
                item = std::move(computed.front());
                computed.pop_front();
            }
            format_slab(fjob, *item.second);
            pool.release(item.second->tallies, item.second->tallies_cap);
            item.second->tallies = nullptr;
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[item.first] = std::move(item.second);
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> workers;
    for (size_t g = 0; g < gpus.size(); ++g)
        workers.emplace_back(gpu_worker, g);
    for (size_t f = 0; f < n_formatters; ++f)
        workers.emplace_back(formatter);
    // ordered writer (the reference's gather_write re-orders by batch idx: src/lib.rs:616-637)
    while (next_to_write < slabs.size()) {
        std::unique_ptr<Slab> s;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[next_to_write] != nullptr; });
            s = std::move(ready[next_to_write]);
        }
        if (s->gtext) {
            wr.write(s->gtext, s->gtext_len);
            pool.release(reinterpret_cast<uint32_t *>(s->gtext), s->gtext_cap);
        }
        for (const TextBuf &part : s->text)
            wr.write(part.p.get(), part.len);
        {
            std::lock_guard<std::mutex> lk(mu);
            ++next_to_write;
        }
        cv.notify_all();
    }
    for (auto &t : workers)
        t.join();
}

// ---------------------------------------------------------------- host self-tests (no GPU) -----
int host_selftest(const Args &a)
{
    if (a.selftest == "fasta") {  // parse stdin, print one line per record
        uint8_t table[256];
        encoding_array(table);
        cli::FastaReader reader(stdin);
        cli::FastaRecord rec;
        for (;;) {
            const int rc = reader.next(rec);
            if (rc < 0) {
                std::printf("ERROR\t%s\n", reader.error().c_str());
                return 0;
            }
            if (rc == 0 || (rec.id.empty() && !rec.has_desc && rec.seq.empty()))
                break;
            std::printf("%s\t%s\t%s\n", rec.id.c_str(), rec.has_desc ? rec.desc.c_str() : "<none>", rec.seq.c_str());
        }
        return 0;
    }
    if (a.selftest == "fasta-blocks") {  // the same through BlockReader + parallel parse, tiny blocks
        uint8_t table[256];
        encoding_array(table);
        std::memset(table, 1, 256);  // accept every byte: this mode checks tokenisation only
        size_t total = 0;
        parse_stream(stdin, a.slab_pairs < 4096 ? a.slab_pairs : 7, 4, table, false, false, 0,
                     [&](std::unique_ptr<Alignment> part) {
                         for (size_t r = 0; r < part->n; ++r)
                             std::printf("%s\t%zu\n", part->ids[r].c_str(), part->width);
                         total += part->n;
                     });
        std::printf("records\t%zu\n", total);
        return 0;
    }
    if (a.selftest == "format") {  // hex floats on stdin -> {:.12}
        char line[256], out[512];
        while (std::fgets(line, sizeof line, stdin)) {
            const double v = std::strtod(line, nullptr);
            const int n = cli::fmt_fixed12(v, out);
            out[n] = 0;
            std::puts(out);
        }
        return 0;
    }
    if (a.selftest == "args") {
        std::printf("measure=%s threads=%zu%s batchsize=%zu stream=%s output=%s gpus=%d inputs=", a.measure.c_str(),
                    a.threads, a.has_threads ? "" : "(default)", a.batchsize, a.has_stream ? a.stream.c_str() : "<none>",
                    a.has_output ? a.output.c_str() : "<stdout>", a.gpus);
        for (auto &s : a.flag_inputs)
            std::printf("[-i %s]", s.c_str());
        for (auto &s : a.pos_inputs)
            std::printf("[pos %s]", s.c_str());
        std::puts("");
        return 0;
    }
    die_usage("unknown --host-selftest");
}

}  // namespace

// DISTANCE_TIMING=1: phase wall times on stderr
struct PhaseTimer {
    bool on = std::getenv("DISTANCE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void mark(const char *what)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[timing] %-28s %8.1f ms (total %8.1f ms)\n", what,
                     std::chrono::duration<double, std::milli>(now - last).count(),
                     std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

// num_cpus::get() (src/lib.rs:262): the CPUs this process may use — its affinity mask, capped by a cgroup CPU
// quota when the container has one (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).
static size_t available_cpus()
{
    size_t n = std::max<unsigned>(1, std::thread::hardware_concurrency());
    auto quota = [](const char *path_quota, const char *path_period) -> double {
        double q = -1, p = 100000;
        if (FILE *fh = std::fopen(path_quota, "r")) {
            char word[64] = {0};
            if (path_period == nullptr) {  // "max 100000" or "1600000 100000"
                double per = 0;
                if (std::fscanf(fh, "%63s %lf", word, &per) == 2 && std::strcmp(word, "max") != 0) {
                    q = std::atof(word);
                    p = per;
                }
            } else if (std::fscanf(fh, "%lf", &q) != 1) {
                q = -1;
            }
            std::fclose(fh);
        }
        if (path_period)
            if (FILE *fh = std::fopen(path_period, "r")) {
                if (std::fscanf(fh, "%lf", &p) != 1)
                    p = 100000;
                std::fclose(fh);
            }
        return (q > 0 && p > 0) ? q / p : -1.0;
    };
    double c = quota("/sys/fs/cgroup/cpu.max", nullptr);
    if (c < 0)
        c = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (c > 0)
        n = std::min<size_t>(n, std::max<size_t>(1, (size_t)(c + 0.999)));
    return n;
}

int main(int argc, char **argv)
{
    PhaseTimer timer;
    std::signal(SIGPIPE, SIG_IGN);  // EPIPE is handled at the write (exit 0), like the reference
    Args a = parse_args(argc, argv);
    if (!a.selftest.empty())
        return host_selftest(a);
    if (a.licenses) {  // src/main.rs:7-10
        std::puts(kLicences);
        return 0;
    }
    // ---- set_up(): src/lib.rs:162-267 -------------------------------------------------------
    if (!a.pos_inputs.empty() && !a.flag_inputs.empty())
        die_message("For loading input files, don't use both positional arguments and the -i/--input flag");
    std::vector<std::string> inputs = a.flag_inputs;
    inputs.insert(inputs.end(), a.pos_inputs.begin(), a.pos_inputs.end());
    std::vector<FILE *> files;
    if (inputs.empty())
        files.push_back(stdin);
    for (auto &p : inputs)
        files.push_back(open_input(p));
    FILE *stream_fh = nullptr;
    if (a.has_stream) {
        if (inputs.size() != 1)
            die_message("If you stream one file, you must also provide exactly one other file to be loaded");
        stream_fh = a.stream == "-" ? stdin : open_input(a.stream);
    }
    // ---- GPUs: the HIP runtime and the contexts come up on a thread of their own while the files are parsed (0.2 s of
    // a 2.3 s run at 50,000 x 30,000); what it finds is looked at after the parse, so a bad input is reported first,
    // as before.  One context (and one worker thread) per listed device; a device may be listed more than once.
    struct GpuInit {
        int ndev = 0;
        bool count_ok = false;
        std::vector<int> devices;
        std::vector<Ctx> gpus;
        std::string error;
    } gi;
    std::thread gpu_init([&gi, &a] {
        gi.count_ok = dst_device_count(&gi.ndev) == DST_OK && gi.ndev > 0;
        if (!gi.count_ok)
            return;
        gi.devices = a.devices;
        if (gi.devices.empty())
            for (int g = 0; g < std::max(1, std::min(a.gpus, gi.ndev)); ++g)
                gi.devices.push_back(g);
        gi.gpus.resize(gi.devices.size());
        for (size_t g = 0; g < gi.devices.size(); ++g)
            if (dst_create(gi.devices[g], &gi.gpus[g].h) != DST_OK) {
                gi.error = dst_last_error(nullptr);
                return;
            }
    });
    uint8_t table[256];
    encoding_array(table);
    const size_t threads = a.has_threads ? std::max<size_t>(a.threads, 1)  // src/lib.rs:253-263
                                         : available_cpus();
    const size_t parse_threads = std::min<size_t>(threads, 16);
    std::vector<Alignment> loaded;
    for (size_t k = 0; k < files.size(); ++k) {  // load_fastas(): src/fastaio.rs:202-212
        loaded.push_back(load_fasta(files[k], table, parse_threads));
        if (k == 1 && loaded[0].width != loaded[1].width)
            die_message(err_lengths(loaded[0].width, loaded[1].width));
    }
    timer.mark("parse + encode loaded files");
    Writer wr;
    if (a.has_output) {
        wr.fh = std::fopen(a.output.c_str(), "wb");
        if (!wr.fh)
            die_io(a.output, errno);
    }
    static char outbuf[1 << 20];
    std::setvbuf(wr.fh, outbuf, _IOFBF, sizeof outbuf);

    const int measure = dst_measure_from_name(a.measure.c_str());
    gpu_init.join();
    if (!gi.count_ok) {
        std::fprintf(stderr, "Error: Gpu(\"no MI355X / HIP device visible: this build has no CPU path\")\n");
        return 1;
    }
    if (!gi.error.empty()) {
        std::fprintf(stderr, "Error: Gpu(\"%s\")\n", gi.error.c_str());
        return 1;
    }
    std::vector<Ctx> &gpus = gi.gpus;
    const int G = (int)gpus.size();
    timer.mark("HIP init + contexts (what the parse did not hide)");
    for (int g = 0; g < G; ++g)
        for (size_t k = 0; k < loaded.size(); ++k)
            gpus[g].check(dst_upload(gpus[g].h, (int)k, loaded[k].codes.data(), loaded[k].n, loaded[k].width,
                                     loaded[k].width, nullptr),
                          "upload");
    // tn93: per-record base counts by code (count_bases(), src/lib.rs:233-239), from the device
    std::vector<std::vector<uint32_t>> counts(loaded.size());
    if (measure == DST_TN93)
        for (size_t k = 0; k < loaded.size(); ++k) {
            counts[k].resize(loaded[k].n * 4);
            gpus[0].check(dst_get_base_counts(gpus[0].h, (int)k, counts[k].data()), "base counts");
        }

    timer.mark("upload + pack + counts");
    static const char header[] = "sequence1\tsequence2\tdistance\n";  // src/lib.rs:613
    wr.write(header, sizeof header - 1);

    Job job;
    job.measure = measure;
    job.fmt_threads = std::max<size_t>(1, threads / (size_t)G);
    if (!stream_fh) {
        // ---- load(): src/lib.rs:367-474 -------------------------------------------------------
        job.square = loaded.size() == 1;
        job.rows = &loaded[0];
        job.cols = &loaded.back();
        job.row_counts = measure == DST_TN93 ? counts[0].data() : nullptr;
        job.col_counts = measure == DST_TN93 ? counts.back().data() : nullptr;
        // The TSV lines themselves come from the GPU (dst_text_*): it has the distances and the ids, the exact
        // {:.12} conversion is integer arithmetic, and the host's formatter pool was what bounded a large run.
        // For jc69 / k80 / tn93 the device finalises the tallies itself and hands the values that lie near a rounding
        // boundary of the 12th decimal back to dst_finalize (libm) before the text leaves the library, so the bytes are
        // the host formatter's (tests/test_gpu_text_identity.py: 0 of 1.5e8 lines differ; DISTANCE_HOST_FORMAT=1 keeps
        // the host formatter, and tests/test_gpu_cli.py compares the two).
        job.gpu_text = std::getenv("DISTANCE_HOST_FORMAT") == nullptr;
        if (job.gpu_text)
            for (int g = 0; g < G && job.gpu_text; ++g)
                for (size_t k = 0; k < loaded.size(); ++k) {
                    std::string chars;
                    std::vector<uint64_t> offs(loaded[k].n + 1, 0);
                    for (size_t r = 0; r < loaded[k].n; ++r) {
                        chars += loaded[k].ids[r];
                        offs[r + 1] = chars.size();
                    }
                    if (dst_set_ids(gpus[g].h, (int)k, chars.data(), offs.data(), loaded[k].n) != DST_OK)
                        job.gpu_text = false;   // (e.g. 4 GB of ids): the host formatter takes over
                }
        run_slabs(gpus, job, 0, 1, a.slab_pairs, wr);
    } else {
        // ---- stream(): src/lib.rs:269-365; stream_fasta(): src/fastaio.rs:215-286 ---------------
        const Alignment &ref = loaded[0];
        // a streamed batch is bounded by records (4096), by result pairs (--slab-pairs) and by bytes
        // (512 MiB of codes: C4-sized 5 Mbp records come ~100 at a time; split-L launches keep the GPU
        // busy on such short batches)
        const size_t by_bytes = ((size_t)512 << 20) / std::max<size_t>(ref.width, 1);
        const size_t batch_records =
            std::max<size_t>(1, std::min<size_t>({(size_t)4096, a.slab_pairs / std::max<size_t>(ref.n, 1), by_bytes}));
        // reader thread: parse + encode batch k+1 while the GPUs and the formatting pool work on
        // batch k (the reference's stream_fasta thread + bounded channel, src/lib.rs:272, 290-307)
        std::mutex qmu;
        std::condition_variable qcv;
        std::deque<std::unique_ptr<Alignment>> queue;
        bool reader_done = false;
        size_t record_counter = 0;
        // bytes of FASTA per parsed block ~ records per batch
        const size_t block_bytes = std::max<size_t>(batch_records * (ref.width + 64), (size_t)1 << 20);
        std::thread reader_thread([&] {
            parse_stream(stream_fh, block_bytes, parse_threads, table, measure == DST_TN93, true, ref.width,
                         [&](std::unique_ptr<Alignment> batch) {
                             if (batch->n == 0)
                                 return;
                             record_counter += batch->n;
                             std::unique_lock<std::mutex> lk(qmu);
                             qcv.wait(lk, [&] { return queue.size() < 2; });
                             queue.push_back(std::move(batch));
                             qcv.notify_all();
                         });
            std::lock_guard<std::mutex> lk(qmu);
            reader_done = true;
            qcv.notify_all();
        });
        // Batches go round-robin to the GPUs (batch b -> GPU b mod G), each GPU runs them through its own
        // overlapped pipeline (dst_stream_*: page-locked ring slots; H2D of the next batch and D2H of the previous
        // one under the compare of the current one), the worker formats what it collects, and the main thread
        // writes the batches strictly in input order (gather_write's idx re-ordering, src/lib.rs:616-637).
        constexpr int kDepth = 3;
        struct Item {
            size_t idx;
            std::unique_ptr<Alignment> batch;
        };
        std::vector<std::deque<Item>> gq((size_t)G);       // per-GPU input queues (guarded by qmu)
        std::vector<bool> gdone((size_t)G, false);
        std::map<size_t, std::vector<TextBuf>> done_text;   // formatted batches waiting for the writer
        std::mutex wmu;
        std::condition_variable wcv;
        size_t next_to_write = 0, n_batches = 0;
        bool all_dispatched = false;
        const size_t window = (size_t)G * (kDepth + 2);     // bound on batches between dispatch and write
        std::vector<dst_stream *> streams((size_t)G, nullptr);
        // what crosses the host link: the codes' high nibbles, two sites per byte (all a measure reads: half the bytes);
        // DISTANCE_WIRE=codes keeps the Paradis bytes (tests run both)
        const char *wire_env = std::getenv("DISTANCE_WIRE");
        const bool nibbles = !(wire_env && std::strcmp(wire_env, "codes") == 0);
        for (int g = 0; g < G; ++g)
            gpus[g].check(dst_stream_open_wire(gpus[g].h, measure, DST_OUT_TALLY, batch_records, kDepth,
                                               nibbles ? DST_WIRE_NIBBLES : DST_WIRE_CODES, &streams[g]), "stream open");
        auto gpu_stream_worker = [&](int g) {
            std::deque<Item> inflight;
            auto collect_one = [&]() {
                size_t n_rec = 0;
                const void *res = nullptr;
                gpus[g].check(dst_stream_collect(streams[g], &n_rec, &res), "stream collect");
                Item it = std::move(inflight.front());
                inflight.pop_front();
                Job sj = job;
                sj.square = false;
                sj.swap_ids = true;          // id1 = loaded record, id2 = streamed record (src/lib.rs:327-330)
                sj.rows = it.batch.get();    // streamed record outer ...
                sj.cols = &ref;              // ... loaded record inner (src/lib.rs:323-324)
                sj.row_counts = measure == DST_TN93 ? it.batch->counts.data() : nullptr;
                sj.col_counts = measure == DST_TN93 ? counts[0].data() : nullptr;
                Slab slab;
                slab.rb = 0;
                slab.re = it.batch->n;
                slab.tallies = const_cast<uint32_t *>(static_cast<const uint32_t *>(res));  // library-owned, read only
                format_slab(sj, slab);
                {
                    std::lock_guard<std::mutex> lk(wmu);
                    done_text[it.idx] = std::move(slab.text);
                }
                wcv.notify_all();
            };
            for (;;) {
                Item it;
                bool have = false;
                {
                    std::unique_lock<std::mutex> lk(qmu);
                    qcv.wait(lk, [&] { return !gq[(size_t)g].empty() || gdone[(size_t)g]; });
                    if (!gq[(size_t)g].empty()) {
                        it = std::move(gq[(size_t)g].front());
                        gq[(size_t)g].pop_front();
                        have = true;
                    }
                }
                qcv.notify_all();
                if (!have)
                    break;
                if (inflight.size() == (size_t)kDepth - 1)
                    collect_one();
                uint8_t *buf = nullptr;
                size_t pitch = 0;
                uint32_t *cbuf = nullptr;
                gpus[g].check(dst_stream_acquire(streams[g], &buf, &pitch, &cbuf), "stream acquire");
                const Alignment &al = *it.batch;
                for (size_t r = 0; r < al.n; ++r) {
                    const uint8_t *src = al.codes.data() + r * al.width;
                    uint8_t *dst = buf + r * pitch;
                    if (!nibbles) {
                        std::memcpy(dst, src, al.width);
                        continue;
                    }
                    const size_t half = al.width / 2;
                    for (size_t k = 0; k < half; ++k)   // site 2k in the low nibble, 2k + 1 in the high one
                        dst[k] = (uint8_t)((src[2 * k] >> 4) | (src[2 * k + 1] & 0xF0u));
                    if (al.width & 1)
                        dst[half] = (uint8_t)((src[al.width - 1] >> 4) | 0xF0u);
                }
                if (measure == DST_TN93)
                    std::memcpy(cbuf, al.counts.data(), al.n * 4 * sizeof(uint32_t));
                gpus[g].check(dst_stream_submit(streams[g], al.n, measure == DST_TN93 ? 1 : 0), "stream submit");
                inflight.push_back(std::move(it));
            }
            while (!inflight.empty())
                collect_one();
        };
        std::vector<std::thread> workers;
        for (int g = 0; g < G; ++g)
            workers.emplace_back(gpu_stream_worker, g);
        // dispatcher: the reader's batches, in order, to the GPUs
        std::thread dispatcher([&] {
            for (;;) {
                std::unique_ptr<Alignment> batch;
                {
                    std::unique_lock<std::mutex> lk(qmu);
                    qcv.wait(lk, [&] { return !queue.empty() || reader_done; });
                    if (queue.empty())
                        break;
                    batch = std::move(queue.front());
                    queue.pop_front();
                }
                qcv.notify_all();
                // a parsed block may hold more records than one pipeline slot: cut it into batches
                for (size_t r0 = 0; r0 < batch->n; r0 += batch_records) {
                    std::unique_ptr<Alignment> piece;
                    if (r0 == 0 && batch->n <= batch_records) {
                        piece = std::move(batch);
                    } else {
                        const size_t r1 = std::min(batch->n, r0 + batch_records);
                        piece = std::make_unique<Alignment>();
                        piece->n = r1 - r0;
                        piece->width = batch->width;
                        piece->ids.assign(batch->ids.begin() + (long)r0, batch->ids.begin() + (long)r1);
                        piece->codes.assign(batch->codes.begin() + (long)(r0 * batch->width),
                                            batch->codes.begin() + (long)(r1 * batch->width));
                        if (!batch->counts.empty())
                            piece->counts.assign(batch->counts.begin() + (long)(4 * r0), batch->counts.begin() + (long)(4 * r1));
                    }
                    size_t idx;
                    {   // keep the number of batches between dispatch and write bounded (memory)
                        std::unique_lock<std::mutex> lk(wmu);
                        wcv.wait(lk, [&] { return n_batches < next_to_write + window; });
                        idx = n_batches++;
                    }
                    {
                        std::lock_guard<std::mutex> lk(qmu);
                        gq[idx % (size_t)G].push_back(Item{idx, std::move(piece)});
                    }
                    qcv.notify_all();
                    if (!batch)
                        break;
                }
            }
            {
                std::lock_guard<std::mutex> lk(qmu);
                for (int g = 0; g < G; ++g)
                    gdone[(size_t)g] = true;
            }
            qcv.notify_all();
            {
                std::lock_guard<std::mutex> lk(wmu);
                all_dispatched = true;
            }
            wcv.notify_all();
        });
        for (;;) {  // ordered writer
            std::vector<TextBuf> text;
            {
                std::unique_lock<std::mutex> lk(wmu);
                wcv.wait(lk, [&] { return done_text.count(next_to_write) || (all_dispatched && next_to_write >= n_batches); });
                if (!done_text.count(next_to_write))
                    break;
                text = std::move(done_text[next_to_write]);
                done_text.erase(next_to_write);
            }
            for (const TextBuf &part : text)
                wr.write(part.p.get(), part.len);
            {
                std::lock_guard<std::mutex> lk(wmu);
                ++next_to_write;
            }
            wcv.notify_all();
        }
        dispatcher.join();
        for (auto &t : workers)
            t.join();
        for (int g = 0; g < G; ++g)
            dst_stream_close(streams[g]);
        reader_thread.join();
        if (record_counter == 0)
            die_message("Empty FASTA file");  // src/fastaio.rs:281-283
    }
    timer.mark("compute + format + write");
    wr.flush();
    timer.mark("flush");
    for (auto &g : gpus)
        dst_destroy(g.h);
    timer.mark("destroy contexts");
    if (a.has_output && std::fclose(wr.fh) != 0)
        die_io(a.output, errno);
    // everything is written and closed: leave without the runtime's own teardown (unlocking the page-locked text buffers,
    // unloading the code objects: 0.3 s of a 2 s run that nothing is waiting for)
    leave(0);
}
