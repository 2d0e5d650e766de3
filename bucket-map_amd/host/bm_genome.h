// bm_genome.h -- reference genome in memory, FASTA/FASTQ I/O and bucket cutting (host plumbing).
//
// Restates, without SeqAn3: seqan3::sequence_file_input on FASTA/FASTQ and
// iterate_through_buckets (bucket_map/utils.h:60-102).
#pragma once

#include "bm_common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string_view>
#include <condition_variable>
#include <fstream>
#include <functional>
#include <stdexcept>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace bm {

// Size of the blocks the FASTA / FASTQ readers pull from the file (BM_IO_BLOCK overrides it: tests use
// tiny blocks to put record boundaries everywhere).
inline size_t io_block_bytes() {
    if (const char *e = std::getenv("BM_IO_BLOCK")) {
        const size_t v = std::strtoull(e, nullptr, 10);
        if (v >= 16) return v;
    }
    return 32u << 20;
}

struct Genome {
    std::vector<std::string> ids;      // full FASTA header lines without '>'
    std::vector<std::string> seqs;     // ACGT only (folded like dna5 -> dna4: everything else -> A)
    // assembly gaps of a synthetic genome as (record, start, length), sorted: in `seqs` they are runs of A (what the
    // folding makes of N); write_fasta prints them as N.  Empty for genomes read from a file.
    std::vector<std::array<uint64_t, 3>> gaps;
    uint64_t total_length() const {
        uint64_t t = 0;
        for (auto &s : seqs) t += s.size();
        return t;
    }
};

// FASTA is read as dna5 and converted to dna4 (utils.h:70,91): ambiguity codes become N, then A.
inline char fold_genome_char(char c) {
    switch (c) {
    case 'A': case 'a': return 'A';
    case 'C': case 'c': return 'C';
    case 'G': case 'g': return 'G';
    case 'T': case 't': case 'U': case 'u': return 'T';
    default: return 'A';
    }
}

// The file is mapped, record starts ('>' at the beginning of a line) are found in one memchr sweep, and the
// records are parsed by a few threads, each into a string reserved to the record's size: a 1.7 Gbp genome has
// 28 M lines, and appending them one by one to growing strings took as long as everything else the tool does.
inline Genome read_fasta(const std::string &path, unsigned threads = 0) {
    static const auto fold_or_skip = [] {             // 0: white space inside a sequence line (dropped)
        std::array<char, 256> t{};
        for (int c = 0; c < 256; c++) t[static_cast<size_t>(c)] = (c == ' ' || c == '\t') ? 0 : fold_genome_char(static_cast<char>(c));
        return t;
    }();
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open FASTA file " + path);
    struct stat st {};
    if (::fstat(fd, &st) != 0) {
        ::close(fd);
        throw std::runtime_error("cannot stat FASTA file " + path);
    }
    const size_t size = static_cast<size_t>(st.st_size);
    Genome g;
    if (size == 0) {
        ::close(fd);
        return g;
    }
    void *map = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) throw std::runtime_error("cannot map FASTA file " + path);
    const char *data = static_cast<const char *>(map);
    // first non-empty line must be a header
    size_t first = 0;
    while (first < size && (data[first] == '\n' || data[first] == '\r')) first++;
    if (first < size && data[first] != '>') {
        ::munmap(map, size);
        throw std::runtime_error("FASTA file " + path + " does not start with '>'");
    }
    std::vector<size_t> starts;                       // offset of every '>' that begins a line
    for (size_t pos = first; pos < size;) {
        starts.push_back(pos);
        const char *p = data + pos + 1;
        for (;;) {
            p = static_cast<const char *>(std::memchr(p, '>', static_cast<size_t>(data + size - p)));
            if (!p || p[-1] == '\n') break;
            p++;
        }
        pos = p ? static_cast<size_t>(p - data) : size;
    }
    const size_t n_rec = starts.size();
    g.ids.resize(n_rec);
    g.seqs.resize(n_rec);
    auto parse = [&](size_t r) {
        const char *p = data + starts[r], *end = data + (r + 1 < n_rec ? starts[r + 1] : size);
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *hend = nl ? nl : end;
        size_t hlen = static_cast<size_t>(hend - (p + 1));
        if (hlen && p[hlen] == '\r') hlen--;
        g.ids[r].assign(p + 1, hlen);
        // per line: the bytes folded (blanks and tabs dropped) straight into a string sized once to the record's bytes --
        // an append and a table look-up per byte were 170 MB/s a thread
        std::string &s = g.seqs[r];
        s.resize(static_cast<size_t>(end - hend));
        char *out = s.data();
        size_t w = 0;
        for (p = nl ? nl + 1 : end; p < end;) {
            nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
            const char *lend = nl ? nl : end;
            size_t len = static_cast<size_t>(lend - p);
            if (len && p[len - 1] == '\r') len--;
            if (!std::memchr(p, ' ', len) && !std::memchr(p, '\t', len)) {
                // compare-and-select on bytes: the compiler turns this loop into 16- or 32-byte vector code
                char *o = out + w;
                for (size_t i = 0; i < len; i++) {
                    const unsigned char u = static_cast<unsigned char>(p[i]) & 0xDFu;   // upper case
                    unsigned char f = 'A';
                    f = u == 'C' ? 'C' : f;
                    f = u == 'G' ? 'G' : f;
                    f = (u == 'T' || u == 'U') ? 'T' : f;
                    o[i] = static_cast<char>(f);
                }
                w += len;
            } else {   // white space inside a sequence line: rare, character by character
                for (size_t i = 0; i < len; i++) {
                    const char c = fold_or_skip[static_cast<unsigned char>(p[i])];
                    out[w] = c;
                    w += c != 0;
                }
            }
            p = nl ? nl + 1 : end;
        }
        s.resize(w);
    };
    if (threads == 0) threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    threads = static_cast<unsigned>(std::min<size_t>(threads, n_rec));
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex err_mu;
    auto work = [&]() {
        try {
            for (size_t r = next.fetch_add(1); r < n_rec; r = next.fetch_add(1)) parse(r);
        } catch (...) {
            std::lock_guard<std::mutex> lk(err_mu);
            if (!err) err = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    ::munmap(map, size);
    if (err) std::rethrow_exception(err);
    return g;
}

inline void write_fasta(const Genome &g, const std::string &path, size_t width = 60) {
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("cannot write FASTA file " + path);
    size_t gi = 0;
    std::string line;
    for (size_t r = 0; r < g.ids.size(); r++) {
        out << '>' << g.ids[r] << '\n';
        const std::string &s = g.seqs[r];
        while (gi < g.gaps.size() && g.gaps[gi][0] < r) gi++;
        for (size_t i = 0; i < s.size(); i += width) {
            const size_t n = std::min(width, s.size() - i);
            while (gi < g.gaps.size() && g.gaps[gi][0] == r && g.gaps[gi][1] + g.gaps[gi][2] <= i) gi++;
            if (gi < g.gaps.size() && g.gaps[gi][0] == r && g.gaps[gi][1] < i + n) {   // a gap touches this line
                line.assign(s.data() + i, n);
                for (size_t k = gi; k < g.gaps.size() && g.gaps[k][0] == r && g.gaps[k][1] < i + n; k++)
                    for (size_t at = std::max<size_t>(i, g.gaps[k][1]); at < std::min<size_t>(i + n, g.gaps[k][1] + g.gaps[k][2]); at++)
                        line[at - i] = 'N';
                out.write(line.data(), static_cast<std::streamsize>(n));
            } else {
                out.write(s.data() + i, static_cast<std::streamsize>(n));
            }
            out.put('\n');
        }
    }
}

// Views into the reader's block buffer: valid only during the callback.
struct FastqRecord {
    std::string_view id;      // header line without '@', untruncated
    std::string_view seq;     // as in the file (ASCII); dna4 folding happens where it is hashed
    std::string_view qual;    // phred+33
};

// for_each_fastq_stream: the portable reader for anything that is not a regular file (a pipe) -- the stream is read in
// blocks and lines are found with memchr; op runs for every record of a 4-line FASTQ stream, in order.  Regular files go
// through FastqFile / for_each_fastq below.
inline void for_each_fastq_stream(const std::string &path, const std::function<void(const FastqRecord &)> &op) {
    // closed on every way out, the callback's exceptions included
    std::unique_ptr<FILE, int (*)(FILE *)> file(std::fopen(path.c_str(), "rb"), &std::fclose);
    if (!file) throw std::runtime_error("cannot open FASTQ file " + path);
    FILE *f = file.get();
    std::vector<char> buf(io_block_bytes());
    size_t have = 0;
    bool eof = false;
    auto fail = [&](const std::string &what) { throw std::runtime_error(what + " in " + path); };
    for (;;) {
        if (!eof) {
            const size_t got = std::fread(buf.data() + have, 1, buf.size() - have, f);
            have += got;
            if (got == 0) eof = true;
        }
        size_t pos = 0;
        for (;;) {
            // four lines; at end of file the last line may lack its '\n'
            const char *line[4];
            size_t len[4];
            size_t p = pos;
            int got_lines = 0;
            while (got_lines < 4) {
                if (p > have) break;
                const char *nl = p < have ? static_cast<const char *>(std::memchr(buf.data() + p, '\n', have - p)) : nullptr;
                size_t end;
                if (nl) end = static_cast<size_t>(nl - buf.data());
                else if (eof && p < have) end = have;
                else if (eof && got_lines == 3 && len[1] == 0) end = have;   // an empty last record may end at its '+' line
                else break;
                line[got_lines] = buf.data() + p;
                len[got_lines] = end - p;
                if (len[got_lines] && line[got_lines][len[got_lines] - 1] == '\r') len[got_lines]--;
                got_lines++;
                p = end + 1;
                if (got_lines == 1 && len[0] == 0) {   // blank line between records: skip it
                    pos = p;
                    got_lines = 0;
                }
            }
            if (got_lines < 4) {
                if (eof && got_lines > 0) fail("truncated FASTQ record");
                break;
            }
            if (line[0][0] != '@') fail("FASTQ record does not start with '@'");
            if (len[2] == 0 || line[2][0] != '+') fail("FASTQ record lacks its '+' line");
            if (len[1] != len[3]) fail("sequence and quality lengths differ");
            FastqRecord rec{std::string_view(line[0] + 1, len[0] - 1), std::string_view(line[1], len[1]),
                            std::string_view(line[3], len[3])};
            op(rec);
            pos = p > have ? have : p;
        }
        if (eof) break;
        // keep the unfinished tail, grow the buffer if one record does not fit
        std::memmove(buf.data(), buf.data() + pos, have - pos);
        have -= pos;
        if (have == buf.size()) buf.resize(buf.size() * 2);
    }
}

namespace fastq_detail {

// [p, line end) of the line that starts at p; next = start of the following line (or end)
inline const char *line_end(const char *p, const char *end, const char *&next) {
    const char *nl = p < end ? static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p))) : nullptr;
    next = nl ? nl + 1 : end;
    const char *e = nl ? nl : end;
    if (e > p && e[-1] == '\r') e--;
    return e;
}

// First record start at or after `from` (a line start): nullptr if there is none before `limit`.
inline const char *find_record(const char *from, const char *limit, const char *end) {
    for (const char *p = from; p < limit;) {
        const char *n1, *n2;
        line_end(p, end, n1);
        if (*p == '@' && n1 < end) {
            line_end(n1, end, n2);
            if (n2 < end && *n2 == '+') return p;
        }
        p = n1;
    }
    return nullptr;
}

// One record as offsets into the mapped file.
struct Rec {
    uint64_t id, seq, qual;     // first byte of the header (after '@'), of the sequence, of the qualities
    uint32_t id_len, len;
};

struct Chunk {
    std::vector<Rec> recs;
    const char *first = nullptr, *stop = nullptr;   // where its first record starts / where parsing stopped
    std::string error;
};

// Records that start in [begin, limit).  `begin` is a line start (or the start of the file).
inline void parse_chunk(const char *data, const char *begin, const char *limit, const char *end, bool is_first, Chunk &out) {
    const char *p = begin;
    if (is_first) {
        while (p < end && (*p == '\n' || *p == '\r')) p++;          // leading blank lines
    } else {
        p = find_record(begin, limit, end);
        if (!p) return;
    }
    out.first = p;
    while (p < limit && p < end) {
        const char *n0, *n1, *n2, *n3;
        const char *e0 = line_end(p, end, n0);
        if (e0 == p) {                                               // blank line between records
            p = n0;
            continue;
        }
        if (*p != '@') {
            out.error = "FASTQ record does not start with '@'";
            return;
        }
        if (n0 >= end) {
            out.error = "truncated FASTQ record";
            return;
        }
        const char *e1 = line_end(n0, end, n1);
        if (n1 >= end) {
            out.error = "truncated FASTQ record";
            return;
        }
        if (*n1 != '+') {                                            // (find_record asks the same of every later chunk)
            out.error = "FASTQ record lacks its '+' line";
            return;
        }
        line_end(n1, end, n2);
        if (n2 >= end && e1 != n0) {                                 // no quality line although there is a sequence
            out.error = "truncated FASTQ record";
            return;
        }
        n3 = end;                                                    // an empty last record may end at its '+' line
        const char *e3 = n2 < end ? line_end(n2, end, n3) : n2;
        if (static_cast<size_t>(e1 - n0) != static_cast<size_t>(e3 - n2)) {
            out.error = "sequence and quality lengths differ";
            return;
        }
        if (static_cast<size_t>(e1 - n0) > 0xFFFFFFFFull || static_cast<size_t>(e0 - p - 1) > 0xFFFFFFFFull) {
            out.error = "FASTQ record longer than 4 Gbases";
            return;
        }
        out.recs.push_back(Rec{static_cast<uint64_t>(p + 1 - data), static_cast<uint64_t>(n0 - data), static_cast<uint64_t>(n2 - data),
                               static_cast<uint32_t>(e0 - p - 1), static_cast<uint32_t>(e1 - n0)});
        p = n3;
    }
    out.stop = p;
}

}  // namespace fastq_detail

// A regular FASTQ file, mapped ONCE and indexed ONCE for every pass over it: the reference walks the file three times
// (mapper, the locator's k-mer sampling, the SAM pass) and so did rounds 1-3 here.  The index (32 bytes a record) is built
// by a thread of its own -- the file is cut into chunks, a few threads find and check the records that START in each (a
// record start is a line that begins with '@' whose line after next begins with '+': a quality line may begin with '@',
// but then the line after next is a sequence, and sequences do not begin with '+') -- and published group by group, so a
// consumer works on the first records while the last are still being found.  Readers only ever see published records.
class FastqFile {
public:
    using Rec = fastq_detail::Rec;
    static constexpr size_t kBlock = 1u << 16;       // records per storage block

    // nullptr when `path` is not a regular file that can be mapped (a pipe: for_each_fastq_stream serves it)
    static std::shared_ptr<FastqFile> open(const std::string &path) {
        struct stat st {};
        if (::stat(path.c_str(), &st) != 0) throw std::runtime_error("cannot open FASTQ file " + path);
        if (!S_ISREG(st.st_mode)) return nullptr;
        // one file is kept: the passes of a run follow each other (or run side by side) on the same file
        static std::mutex mu;
        static std::shared_ptr<FastqFile> last;
        std::lock_guard<std::mutex> lock(mu);
        if (last && last->path_ == path && last->dev_ == st.st_dev && last->ino_ == st.st_ino &&
            last->size_ == static_cast<size_t>(st.st_size) && last->mtime_ns_ == mtime_ns(st))
            return last;
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open FASTQ file " + path);
        const size_t size = static_cast<size_t>(st.st_size);
        void *map = size ? ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        ::close(fd);
        if (map == MAP_FAILED) return nullptr;                       // no address space: the block reader
        std::shared_ptr<FastqFile> f(new FastqFile());
        f->path_ = path;
        f->dev_ = st.st_dev;
        f->ino_ = st.st_ino;
        f->mtime_ns_ = mtime_ns(st);
        f->data_ = static_cast<const char *>(map);
        f->size_ = size;
        if (size) ::madvise(map, size, MADV_SEQUENTIAL);
        f->blocks_.resize(size / (6 * kBlock) + 2);                  // a record is at least "@\n\n+\n\n"
        f->builder_ = std::thread([raw = f.get()]() { raw->build(); });
        last = f;
        return f;
    }

    ~FastqFile() {
        if (builder_.joinable()) builder_.join();
        if (data_ && size_) ::munmap(const_cast<char *>(data_), size_);
    }
    FastqFile(const FastqFile &) = delete;
    FastqFile &operator=(const FastqFile &) = delete;

    const char *data() const { return data_; }
    size_t size() const { return size_; }
    const std::string &path() const { return path_; }

    // Blocks until `want` records are published or the whole file is indexed; returns how many are (never fewer than an
    // earlier call returned).  Throws the file's parse error once every record before it has been handed out.
    size_t wait(size_t want) {
        size_t n = ready_.load(std::memory_order_acquire);
        if (n >= want) return n;
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&]() { return ready_.load(std::memory_order_acquire) >= want || done_; });
        n = ready_.load(std::memory_order_acquire);
        if (n < want && !error_.empty()) throw std::runtime_error(error_ + " in " + path_);
        return n;
    }
    size_t count() { return wait(~static_cast<size_t>(0)); }         // all of them (waits for the end of the file)

    const Rec &rec(size_t i) const { return blocks_[i / kBlock][i % kBlock]; }
    FastqRecord view(size_t i) const {
        const Rec &r = rec(i);
        return FastqRecord{std::string_view(data_ + r.id, r.id_len), std::string_view(data_ + r.seq, r.len),
                           std::string_view(data_ + r.qual, r.len)};
    }

private:
    FastqFile() = default;
    static int64_t mtime_ns(const struct stat &st) { return static_cast<int64_t>(st.st_mtim.tv_sec) * 1000000000ll + st.st_mtim.tv_nsec; }

    void publish(size_t n, bool done, std::string error = {}) {
        std::lock_guard<std::mutex> lock(mu_);
        ready_.store(n, std::memory_order_release);
        if (done) {
            error_ = std::move(error);
            done_ = true;
        }
        cv_.notify_all();
    }

    void build() {
        using fastq_detail::Chunk;
        const char *data = data_, *end = data_ + size_;
        const size_t chunk = std::min<size_t>(io_block_bytes(), 4u << 20), n_chunks = (size_ + chunk - 1) / chunk;
        const unsigned threads = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
        const size_t group = std::max<size_t>(1, std::min<size_t>(threads, n_chunks));
        size_t n = 0;
        try {
            // chunk c covers [c * chunk, (c + 1) * chunk), moved forward to the next line start
            auto parse_group = [&](size_t g0, std::vector<Chunk> &out) {
                const size_t cnt = std::min(group, n_chunks - g0);
                out.assign(cnt, Chunk());
                std::atomic<size_t> next{0};
                auto work = [&]() {
                    for (size_t i = next.fetch_add(1); i < cnt; i = next.fetch_add(1)) {
                        const size_t c = g0 + i;
                        const char *b = data + c * chunk, *limit = std::min(end, data + (c + 1) * chunk);
                        if (c > 0 && b[-1] != '\n') {                // inside a line: this chunk starts at the next one
                            const char *nl = static_cast<const char *>(std::memchr(b, '\n', static_cast<size_t>(end - b)));
                            b = nl ? nl + 1 : end;
                        }
                        if (b < limit || c == 0) fastq_detail::parse_chunk(data, b, limit, end, c == 0, out[i]);
                    }
                };
                std::vector<std::thread> pool;
                for (size_t t = 1; t < std::min<size_t>(threads, cnt); t++) pool.emplace_back(work);
                work();
                for (auto &t : pool) t.join();
            };
            std::vector<Chunk> cur;
            const char *expect = nullptr;                            // where the next record must start
            for (size_t g0 = 0; g0 < n_chunks; g0 += group) {
                parse_group(g0, cur);
                for (Chunk &c : cur) {
                    if (c.first) {
                        // every byte between two records was looked at: the chunks' records join up exactly (blank lines aside)
                        if (expect) {
                            const char *q = expect;
                            while (q < c.first && (*q == '\n' || *q == '\r')) q++;
                            if (q != c.first) return publish(n, true, "malformed FASTQ record");
                        }
                        for (size_t at = 0; at < c.recs.size();) {   // into the storage blocks
                            const size_t blk = n / kBlock, off = n % kBlock, take = std::min(c.recs.size() - at, kBlock - off);
                            if (blk >= blocks_.size()) return publish(n, true, "more FASTQ records than the file can hold");
                            if (!blocks_[blk]) blocks_[blk].reset(new Rec[kBlock]);
                            std::memcpy(blocks_[blk].get() + off, c.recs.data() + at, take * sizeof(Rec));
                            at += take;
                            n += take;
                        }
                        expect = c.stop;
                    }
                    if (!c.error.empty()) return publish(n, true, c.error);
                }
                publish(n, false);
            }
            // whatever follows the last record must be blank
            for (const char *q = expect ? expect : data; q < end; q++)
                if (*q != '\n' && *q != '\r') return publish(n, true, "truncated FASTQ record");
            publish(n, true);
        } catch (const std::exception &e) {
            publish(n, true, std::string("FASTQ index: ") + e.what());
        }
    }

    std::string path_;
    dev_t dev_ = 0;
    ino_t ino_ = 0;
    int64_t mtime_ns_ = 0;
    const char *data_ = nullptr;
    size_t size_ = 0;
    std::vector<std::unique_ptr<Rec[]>> blocks_;
    std::atomic<size_t> ready_{0};
    bool done_ = false;
    std::string error_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread builder_;
};

// Calls op for every record of a 4-line FASTQ file, in file order, on the calling thread (views into the mapped file: valid
// while the FastqFile lives -- at least for the duration of the call).
inline void for_each_fastq(const std::string &path, const std::function<void(const FastqRecord &)> &op) {
    std::shared_ptr<FastqFile> f = FastqFile::open(path);
    if (!f) {                                                        // a pipe, or no address space: opened ONCE, read in blocks
        for_each_fastq_stream(path, op);
        return;
    }
    for (size_t i = 0;;) {
        const size_t n = f->wait(i + 1);
        if (n <= i) break;
        for (; i < n; i++) op(f->view(i));
    }
}

// One kept bucket of iterate_through_buckets (utils.h:72-97).
struct Bucket {
    uint32_t record;     // index of the FASTA record
    uint32_t index;      // i: position of the bucket inside the record (kept AND skipped count)
    uint32_t start, end; // [start, end) in the record; end - start > read_len
};

// utils.h:72-97.  n = ceil(float(len)/bucket_len) in FLOAT32; bucket i = [i*bl, min(i*bl+bl+rl, len));
// buckets with size <= read_len are skipped; global id = running count of kept buckets.
inline std::vector<Bucket> cut_buckets(const Genome &g, int bucket_length, int read_length) {
    std::vector<Bucket> out;
    for (size_t r = 0; r < g.seqs.size(); r++) {
        const int size = static_cast<int>(g.seqs[r].size());
        volatile float total_length = static_cast<float>(size);
        volatile float quot = total_length / static_cast<float>(bucket_length);
        const int num_buckets = static_cast<int>(std::ceil(static_cast<double>(quot)));
        for (int i = 0; i < num_buckets; i++) {
            int start = i * bucket_length;
            int end = start + bucket_length + read_length;
            if (end > size) end = size;
            if (end - start <= read_length) continue;
            out.push_back(Bucket{static_cast<uint32_t>(r), static_cast<uint32_t>(i), static_cast<uint32_t>(start),
                                 static_cast<uint32_t>(end)});
        }
    }
    return out;
}

// The compile-time BM_BUCKET_NUM of the reference: sum over records of ceil(len/bucket_len) in awk
// double arithmetic (bucket_map/CMakeLists.txt:13-46).  Can exceed the number of kept buckets.
inline uint32_t awk_bucket_num(const Genome &g, uint32_t bucket_length) {
    uint64_t n = 0;
    for (auto &s : g.seqs)
        if (!s.empty()) {
            double v = static_cast<double>(s.size()) / bucket_length;
            n += (v == std::floor(v)) ? static_cast<uint64_t>(v) : static_cast<uint64_t>(v) + 1;
        }
    return static_cast<uint32_t>(n);
}

}  // namespace bm
