// bucket_locator.h -- exact position inside the candidate buckets + SAM output.
//
// Restates bucket_locator (bucket_map/locator/bucket_locator.h) without SeqAn3.  Host side:
//   query_sequences_storage / _prepare_read_query   :19-103, :292-347
//   _locate: bucket loop and its ordering contract   :613-705
//   _filter_best_locations                           :350-405
//   locate (SAM)                                     :455-611  (both branches: with a verifier plugged
//                                                               in, the BM_ALIGN one)
// The candidate scan itself -- _create_kmer_index (:162-177) + _find_offset (:209-290) for every
// candidate (window, bucket, strand) -- sits behind bm::offset_scanner: the MI355X scan (include/bml.h)
// in the product, the C oracle in the test build.  Order-sensitive details are kept on purpose
// (SURVEY App. A.7): revcomp candidates of a bucket are visited in reverse list order, `offset > 0`
// drops an exact hit at bucket offset 0, results are appended per read in bucket order.
#pragma once

#include "bm_genome.h"
#include "mapper.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <charconv>
#include <chrono>
#include <iostream>
#include <memory>
#include <string_view>
#include <thread>
#include <tuple>

namespace bm {

// Where _prepare_read_query's sampling and _create_kmer_index + _find_offset run.  One scan call handles
// every candidate of a _locate pass; sampling is called per block of reads.
class offset_scanner {
public:
    virtual ~offset_scanner() = default;
    // _prepare_read_query (:292-347) for windows given as views into bases / quals: p (hash, position) pairs
    // per window, has[w] = 0 for a window shorter than k
    virtual void sample_windows(const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes, const uint64_t *win_start,
                                const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality,
                                uint32_t *out_hash, uint16_t *out_pos, uint8_t *out_has) = 0;
    // the same for windows whose bases and qualities lie apart in one buffer (the mapped FASTQ file): window w =
    // text[seq_start[w], +win_len[w]) with qualities text[qual_start[w], +win_len[w]).  Default: gather, then sample_windows.
    virtual void sample_text_windows(const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start, const uint64_t *qual_start,
                                     const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality, uint32_t *out_hash,
                                     uint16_t *out_pos, uint8_t *out_has) {
        (void)n_bytes;
        std::vector<uint64_t> start(n_windows);
        uint64_t at = 0;
        for (uint32_t w = 0; w < n_windows; w++) {
            start[w] = at;
            at += win_len[w];
        }
        std::vector<uint8_t> bases(at + 1), quals(at + 1);
        for (uint32_t w = 0; w < n_windows; w++) {
            std::memcpy(bases.data() + start[w], text + seq_start[w], win_len[w]);
            std::memcpy(quals.data() + start[w], text + qual_start[w], win_len[w]);
        }
        sample_windows(bases.data(), quals.data(), at, start.data(), win_len, n_windows, min_base_quality, out_hash, out_pos, out_has);
    }
    // genome as one byte string; bucket b = [bucket_start[b], +bucket_len[b])
    virtual void load_genome(const uint8_t *bases, uint64_t n_bases, const uint64_t *bucket_start,
                             const uint32_t *bucket_len, uint32_t n_buckets) = 0;
    // ... or from records that are buffers of their own (concatenated in order; bucket_start counts in that concatenation).
    // Default: flatten, then load_genome.
    virtual void load_genome_records(const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records,
                                     const uint64_t *bucket_start, const uint32_t *bucket_len, uint32_t n_buckets) {
        uint64_t total = 0;
        for (uint32_t r = 0; r < n_records; r++) total += rec_len[r];
        std::unique_ptr<uint8_t[]> flat(new uint8_t[total ? total : 1]);
        uint64_t at = 0;
        for (uint32_t r = 0; r < n_records; r++) {
            std::memcpy(flat.get() + at, rec[r], rec_len[r]);
            at += rec_len[r];
        }
        load_genome(flat.get(), total, bucket_start, bucket_len, n_buckets);
    }
    // windows: sample_hash/sample_pos [n_windows x p], seg_len[n_windows]; candidates grouped by bucket.
    // out_offset = what _find_offset returns (first of the pair, or -1), out_votes = second.
    virtual void scan(const uint32_t *sample_hash, const uint16_t *sample_pos, const uint32_t *seg_len,
                      uint32_t n_windows, const uint32_t *pair_bucket, const uint32_t *pair_window,
                      const uint8_t *pair_rc, uint32_t n_pairs, int32_t *out_offset, uint32_t *out_votes) = 0;
};

// Where the BM_ALIGN branch's align_pairwise (:520-528,569) runs: the MI355X verifier (include/bmv.h) in the
// product, the C oracle in the test build.  One call handles a batch of (text window, query) pairs given as
// views into the genome string and into a buffer of reads.
class alignment_verifier {
public:
    virtual ~alignment_verifier() = default;
    virtual void load_genome(const uint8_t *bases, uint64_t n_bases) = 0;
    virtual void load_genome_records(const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records) {
        uint64_t total = 0;
        for (uint32_t r = 0; r < n_records; r++) total += rec_len[r];
        std::unique_ptr<uint8_t[]> flat(new uint8_t[total ? total : 1]);
        uint64_t at = 0;
        for (uint32_t r = 0; r < n_records; r++) {
            std::memcpy(flat.get() + at, rec[r], rec_len[r]);
            at += rec_len[r];
        }
        load_genome(flat.get(), total);
    }
    // score = alignment.score(), begin = sequence1_begin_position(), CIGAR entries packed len << 4 | op
    // (0 M, 1 I, 2 D); alignment a owns cigar[cigar_offset[a] .. cigar_offset[a + 1])
    virtual void align(const uint8_t *reads, uint64_t n_read_bytes, const uint64_t *text_start, const uint32_t *text_len,
                       const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len, uint32_t n,
                       std::vector<int32_t> &score, std::vector<uint32_t> &begin, std::vector<uint64_t> &cigar_offset,
                       std::vector<uint32_t> &cigar) = 0;
};

// SAM text as seqan3::sam_file_output lays it out: records are appended to one buffer with
// std::to_chars and go to the file a few megabytes at a time (a million `ostream <<` chains were the slowest stage
// of the tool after the device work had shrunk to milliseconds).
class sam_text {
    std::ofstream out_;
    std::string buf_, writing_;     // the buffer being filled / the one a writer thread is putting into the file
    std::thread writer_;
    bool write_failed_ = false;

    void join_writer() {
        if (writer_.joinable()) writer_.join();
        if (write_failed_) throw std::runtime_error("writing the SAM file failed");
    }

    static void number(std::string &buf, uint64_t v) {
        char tmp[24];
        const auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
        buf.append(tmp, static_cast<size_t>(r.ptr - tmp));
    }
    void number(uint64_t v) { number(buf_, v); }

public:
    explicit sam_text(std::filesystem::path const &file) : out_(file, std::ios::binary) {
        if (!out_) throw std::runtime_error("cannot write " + file.string());
        buf_.reserve(5u << 20);
        writing_.reserve(5u << 20);
    }
    ~sam_text() {
        try {
            flush();
            join_writer();
        } catch (...) {
        }
    }
    // the filled buffer goes to the file on a thread of its own while the next one is filled (records are formatted at
    // about the speed the file system takes them: one after the other they cost twice)
    void flush() {
        join_writer();
        writing_.swap(buf_);
        buf_.clear();
        writer_ = std::thread([this]() {
            out_.write(writing_.data(), static_cast<std::streamsize>(writing_.size()));
            if (!out_) write_failed_ = true;
        });
    }
    // everything written (call before reading the file back or reporting success)
    void close() {
        flush();
        join_writer();
        out_.flush();
        if (!out_) throw std::runtime_error("writing the SAM file failed");
    }
    void header(const std::vector<std::string> &ref_ids, const std::vector<size_t> &ref_lengths) {
        buf_ += "@HD\tVN:1.6\n";
        for (size_t i = 0; i < ref_ids.size(); i++) {
            buf_ += "@SQ\tSN:";
            buf_ += ref_ids[i];
            buf_ += "\tLN:";
            number(ref_lengths[i]);
            buf_ += '\n';
        }
    }
    // QNAME FLAG RNAME POS MAPQ CIGAR * 0 0 SEQ QUAL
    static void format(std::string &buf, std::string_view qname, unsigned flag, std::string_view rname, uint64_t pos, unsigned mapq,
                       std::string_view cigar, std::string_view seq, std::string_view qual) {
        buf.append(qname); buf += '\t';
        number(buf, flag); buf += '\t';
        buf.append(rname); buf += '\t';
        number(buf, pos); buf += '\t';
        number(buf, mapq); buf += '\t';
        buf.append(cigar);
        buf += "\t*\t0\t0\t";
        buf.append(seq); buf += '\t';
        buf.append(qual); buf += '\n';
    }
    void record(std::string_view qname, unsigned flag, std::string_view rname, uint64_t pos, unsigned mapq, std::string_view cigar,
                std::string_view seq, std::string_view qual) {
        format(buf_, qname, flag, rname, pos, mapq, cigar, seq, qual);
        if (buf_.size() > (4u << 20)) flush();
    }
    // records formatted elsewhere (format(): a few threads, each a block of reads), appended in order
    void append(std::string_view records) {
        buf_.append(records);
        if (buf_.size() > (4u << 20)) flush();
    }
};

class bucket_locator {
public:
    // (bucket id, offset in the bucket, window offset in the read, votes, true = read as-is)
    using locate_t = std::tuple<unsigned int, int, unsigned int, unsigned int, bool>;

private:
    mapper *_m;
    offset_scanner *_s;
    alignment_verifier *_v = nullptr;            // non-null: the BM_ALIGN behaviour
    const Genome *genome_ = nullptr;
    std::vector<Bucket> buckets_;
    std::vector<uint64_t> bstart_;               // bucket views into the records laid back to back
    std::vector<uint32_t> blen_;

    unsigned int bucket_length, read_length, min_base_quality;
    uint8_t k;
    int allowed_mismatch, allowed_indel;
    float allowed_indel_rate;
    int num_samples;
    unsigned int num_segment_samples;

    // query_sequences_storage (:19-103), flat: window w of read r = first_window[r] + its rank among the
    // read's windows; samples are num_samples (hash, position) pairs per window.
    std::vector<uint32_t> first_window;          // per read (+1 sentinel)
    std::vector<int> window_start;               // per window: start inside the read
    std::vector<uint32_t> sample_hash;           // per window x num_samples
    std::vector<uint16_t> sample_pos;            // per window x num_samples
    std::vector<uint32_t> segment_length;        // per window
    std::vector<uint8_t> window_has_samples;     // windows shorter than k have none
    std::vector<unsigned int> read_lengths;      // per read

    static uint32_t kmer_hash_at(const char *s, uint32_t k_) {
        uint32_t h = 0;
        for (uint32_t t = 0; t < k_; t++) h = (h << 2) | dna4_rank(static_cast<uint8_t>(s[t]));
        return h;
    }

    uint32_t window_of(const segment_info_t &seg) const {
        for (uint32_t w = first_window[seg.first]; w < first_window[seg.first + 1]; w++)
            if (window_start[w] == seg.second) return w;
        throw std::runtime_error("the mapper returned a (read, window) pair the locator never sampled");
    }

    // _prepare_read_query (:292-347): the FASTQ pass and the windowing (:303-316) stay here; the sampling of
    // every window (:317-343) runs behind offset_scanner::sample_windows, a block of reads at a time.
    void prepare_read_query(const std::string &fastq) {
        first_window.clear(); window_start.clear(); sample_hash.clear(); sample_pos.clear();
        segment_length.clear(); window_has_samples.clear(); read_lengths.clear();
        std::vector<uint8_t> bases, quals;
        std::vector<uint64_t> win_start, qual_start;
        std::vector<uint32_t> win_len;
        const size_t block_bases = 256u << 20;
        // A regular file: nothing is copied here -- the windows are views into the mapped file (the index the mapper's pass
        // built, or is building), and the scanner gathers them where it needs them.
        if (std::shared_ptr<FastqFile> fq = FastqFile::open(fastq)) {
            const uint8_t *text = reinterpret_cast<const uint8_t *>(fq->data());
            size_t block = 0;
            auto flush_views = [&]() {
                const size_t n = win_start.size(), at = sample_hash.size();
                if (n == 0) return;
                sample_hash.resize(at + n * num_samples);
                sample_pos.resize(at + n * num_samples);
                const size_t has_at = window_has_samples.size();
                window_has_samples.resize(has_at + n);
                _s->sample_text_windows(text, fq->size(), win_start.data(), qual_start.data(), win_len.data(), static_cast<uint32_t>(n),
                                        min_base_quality, sample_hash.data() + at, sample_pos.data() + at, window_has_samples.data() + has_at);
                win_start.clear(); qual_start.clear(); win_len.clear();
                block = 0;
            };
            for (size_t i = 0;;) {
                const size_t n = fq->wait(i + 1);
                if (n <= i) break;
                for (; i < n; i++) {
                    const FastqFile::Rec &r = fq->rec(i);
                    const uint32_t len = r.len;
                    first_window.push_back(static_cast<uint32_t>(window_start.size()));
                    auto add = [&](uint32_t st) {
                        const uint32_t end = std::min(st + read_length, len);
                        window_start.push_back(static_cast<int>(st));
                        segment_length.push_back(end - st);
                        win_start.push_back(r.seq + st);
                        qual_start.push_back(r.qual + st);
                        win_len.push_back(end - st);
                    };
                    if (len > 2 * read_length)
                        for (uint32_t st : sample_deterministically(num_segment_samples, len - read_length - 1)) add(st);
                    else
                        add(0);
                    read_lengths.push_back(len);
                    block += len;
                    if (block >= block_bases) flush_views();
                }
            }
            flush_views();
            first_window.push_back(static_cast<uint32_t>(window_start.size()));
            return;
        }
        auto flush = [&]() {
            const size_t n = win_start.size(), at = sample_hash.size();
            if (n == 0) return;
            sample_hash.resize(at + n * num_samples);
            sample_pos.resize(at + n * num_samples);
            const size_t has_at = window_has_samples.size();
            window_has_samples.resize(has_at + n);
            _s->sample_windows(bases.data(), quals.data(), bases.size(), win_start.data(), win_len.data(),
                               static_cast<uint32_t>(n), min_base_quality, sample_hash.data() + at, sample_pos.data() + at,
                               window_has_samples.data() + has_at);
            bases.clear(); quals.clear(); win_start.clear(); win_len.clear();
        };
        for_each_fastq_stream(fastq, [&](const FastqRecord &rec) {
            const uint32_t len = static_cast<uint32_t>(rec.seq.size());
            first_window.push_back(static_cast<uint32_t>(window_start.size()));
            std::vector<uint32_t> starting_positions{0};
            if (len > 2 * read_length) starting_positions = sample_deterministically(num_segment_samples, len - read_length - 1);
            const uint64_t at = bases.size();
            bases.insert(bases.end(), rec.seq.begin(), rec.seq.end());
            quals.insert(quals.end(), rec.qual.begin(), rec.qual.end());
            for (uint32_t i : starting_positions) {
                const uint32_t end = std::min(i + read_length, len);
                window_start.push_back(static_cast<int>(i));
                segment_length.push_back(end - i);
                win_start.push_back(at + i);
                win_len.push_back(end - i);
            }
            read_lengths.push_back(len);
            if (bases.size() >= block_bases) flush();
        });
        flush();
        first_window.push_back(static_cast<uint32_t>(window_start.size()));
    }

    // _filter_best_locations (:350-405).  The reference keeps a std::map keyed (bucket, offset, strand); here the
    // proposals of a read live in one flat vector kept in that key order (a read has a handful of them):
    //   * an incoming location adds its votes to EVERY proposal of its bucket and strand whose offset lies in
    //     [offset - len * n, offset + len * n] -- float32 arithmetic, truncated towards zero (:365-366);
    //   * only when none took the votes does it become a proposal itself, with its own votes (:380);
    //   * all proposals holding the maximum come out, in key order (:390-402).
    // Pinned by tests/golden/sam_small.json (an independent plain-Python statement of the whole tool).
    struct proposal {
        unsigned int bucket;
        int offset;
        bool as_is;
        unsigned int votes;
        bool before(unsigned int b, int o, bool s) const {
            return std::tie(bucket, offset, as_is) < std::tie(b, o, s);
        }
    };
    std::vector<locate_t> filter_best_locations(const std::vector<locate_t> &mapped_locations, unsigned int read_len) const {
        std::vector<proposal> props;
        const float reach = read_len * allowed_indel_rate;
        for (const locate_t &loc : mapped_locations) {
            const unsigned int bucket = std::get<0>(loc), votes = std::get<3>(loc);
            const int offset = std::get<1>(loc);
            const bool as_is = std::get<4>(loc);
            const int lo = static_cast<int>(offset - reach), hi = static_cast<int>(offset + reach);
            bool taken = false;
            for (proposal &p : props)
                if (p.bucket == bucket && p.as_is == as_is && p.offset >= lo && p.offset <= hi) {
                    p.votes += votes;
                    taken = true;
                }
            if (taken) continue;
            auto at = std::find_if(props.begin(), props.end(), [&](const proposal &p) { return !p.before(bucket, offset, as_is); });
            props.insert(at, proposal{bucket, offset, as_is, votes});
        }
        unsigned int top = 0;
        for (const proposal &p : props) top = std::max(top, p.votes);
        std::vector<locate_t> best;
        for (const proposal &p : props)
            if (p.votes == top) best.emplace_back(p.bucket, p.offset, 0u, p.votes, p.as_is);
        return best;
    }

    // SEQ as seqan3 writes a dna4 vector: the reads were folded to A/C/G/T when they were parsed
    // (_phred94_traits, utils.h:192-204), so N / IUPAC / lower case never reach the SAM file.
    static void append_dna4(std::string &out, std::string_view seq) {
        static const std::array<char, 256> fold = [] {
            std::array<char, 256> t{};
            for (int c = 0; c < 256; c++) t[static_cast<size_t>(c)] = dna4_char(dna4_rank(static_cast<uint8_t>(c)));
            return t;
        }();
        const size_t at = out.size();
        out.resize(at + seq.size());
        for (size_t i = 0; i < seq.size(); i++) out[at + i] = fold[static_cast<uint8_t>(seq[i])];
    }

public:
    // bucket_locator ctor (:409-432); the indexer pointer of the reference is replaced by the genome
    // (locator::initialize -> indexer::index is done by the caller, see main.cpp).
    bucket_locator(mapper *map, offset_scanner *scanner, unsigned int bucket_len, unsigned int read_len,
                   uint8_t seed_len, float mismatch_rate, float indel_rate, unsigned int sample_size,
                   unsigned int quality_threshold, unsigned int num_segment_samples_ = 5)
        : _m(map), _s(scanner), bucket_length(bucket_len), read_length(read_len), k(seed_len) {
        allowed_mismatch = static_cast<int>(ceil_mul_f32(mismatch_rate, sample_size));   // :419
        allowed_indel = static_cast<int>(ceil_mul_f32(indel_rate, read_len));            // :420
        allowed_indel_rate = indel_rate;
        num_samples = static_cast<int>(sample_size);
        num_segment_samples = num_segment_samples_;
        min_base_quality = quality_threshold * k;                                         // :431
    }

    // bucketmap_align: every located candidate is verified by a pairwise alignment before it is written
    void set_verifier(alignment_verifier *v) { _v = v; }

    int get_allowed_mismatch() const { return allowed_mismatch; }
    int get_allowed_indel() const { return allowed_indel; }

    // initialize (:440-453): remember the genome, load the q-gram index into the mapper.
    // (Tried: the genome upload begun HERE, under the index load -- everything the locator needs was there no sooner, the
    // upload being the longest of the three passes either way, and map() took twice as long for sharing the link with it
    // from its first batch on: 0.09 -> 0.22 s per 1 M reads.  It starts with map(), in locate_reads.)
    void initialize(const Genome &genome, std::filesystem::path const &index_directory, std::string const &indicator) {
        genome_ = &genome;
        _m->load(index_directory, indicator);
    }

    ~bucket_locator() {
        if (uploader_.joinable()) uploader_.join();
    }

private:
    // BM_SERIAL_PASSES=1 (measurement): the upload and the sampling pass start when map() has returned, so that map()'s own
    // time can be read without two other passes sharing the PCIe link and the cores with it
    bool serial_passes_ = std::getenv("BM_SERIAL_PASSES") != nullptr;
    std::thread uploader_;
    std::exception_ptr upload_error_;
    float upload_ms_ = 0.f;
    bool upload_started_ = false;

    // the bucket sequences as views into one byte string that goes to the devices (_initialize_kmer_index, :151-160)
    void start_genome_upload() {
        if (upload_started_) return;
        upload_started_ = true;
        buckets_ = cut_buckets(*genome_, static_cast<int>(bucket_length), static_cast<int>(read_length));
        const auto t_begin = std::chrono::steady_clock::now();
        uploader_ = std::thread([this, t_begin]() {
            try {
                // the records go to the devices back to back, each from where it lies (no flattened copy on the host)
                const size_t n_rec = genome_->seqs.size();
                std::vector<uint64_t> rec_off(n_rec + 1, 0), rec_len(n_rec);
                std::vector<const uint8_t *> rec(n_rec);
                for (size_t r = 0; r < n_rec; r++) {
                    rec[r] = reinterpret_cast<const uint8_t *>(genome_->seqs[r].data());
                    rec_len[r] = genome_->seqs[r].size();
                    rec_off[r + 1] = rec_off[r] + rec_len[r];
                }
                bstart_.assign(buckets_.size(), 0);
                blen_.assign(buckets_.size(), 0);
                for (size_t b = 0; b < buckets_.size(); b++) {
                    bstart_[b] = rec_off[buckets_[b].record] + buckets_[b].start;
                    blen_[b] = buckets_[b].end - buckets_[b].start;
                }
                _s->load_genome_records(rec.data(), rec_len.data(), static_cast<uint32_t>(n_rec), bstart_.data(), blen_.data(),
                                        static_cast<uint32_t>(buckets_.size()));
                if (_v) _v->load_genome_records(rec.data(), rec_len.data(), static_cast<uint32_t>(n_rec));
            } catch (...) {
                upload_error_ = std::current_exception();
            }
            upload_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        });
    }

public:
    // _locate (:613-705)
    std::vector<std::vector<locate_t>> locate_reads(const std::string &sequence_file) {
        // _initialize_kmer_index (:151-160: the bucket sequences to the devices) and _prepare_read_query (:292-347: the
        // locator's own pass over the FASTQ file) do not depend on the mapper's results, so each runs on a thread of its own
        // while _m->map() lays out the reads' windows and drives the filter.
        if (!serial_passes_) start_genome_upload();
        // (two threads: the upload touches the scanner's genome buffers, the sampling its window buffers and its stream)
        std::exception_ptr sampling_error;
        const auto t_side = std::chrono::steady_clock::now();
        float sampling_ms = 0.f;
        segments_t sequence_ids_orig, sequence_ids_rev_comp;
        if (serial_passes_) {
            std::tie(sequence_ids_orig, sequence_ids_rev_comp) = _m->map(sequence_file);
            _m->reset();
            start_genome_upload();
        }
        std::thread sampler([&]() {
            try {
                prepare_read_query(sequence_file);
            } catch (...) {
                sampling_error = std::current_exception();
            }
            sampling_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_side).count();
        });
        try {
            if (!serial_passes_) {
                std::tie(sequence_ids_orig, sequence_ids_rev_comp) = _m->map(sequence_file);
                _m->reset();
            }
        } catch (...) {
            if (uploader_.joinable()) uploader_.join();
            sampler.join();
            throw;
        }
        auto t0 = std::chrono::steady_clock::now();
        if (uploader_.joinable()) uploader_.join();
        sampler.join();
        upload_started_ = false;                                 // (a second locate() on this object uploads again)
        if (upload_error_) {
            std::exception_ptr e = upload_error_;
            upload_error_ = nullptr;
            std::rethrow_exception(e);
        }
        if (sampling_error) std::rethrow_exception(sampling_error);
        if (std::getenv("BM_LOG_BATCHES"))
            std::cerr << "[bm] beside map(): genome upload done after " << upload_ms_ << " ms, k-mer sampling pass after "
                      << sampling_ms << " ms, map() returned after " << std::chrono::duration<float, std::milli>(t0 - t_side).count()
                      << " ms, both passes and the upload done after "
                      << std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_side).count() << " ms\n";

        // Candidates in the order of the reference's bucket loop (:651-693): buckets ascending; inside a
        // bucket the reads as-is in list order, then the reverse complements in REVERSE list order.
        std::vector<uint32_t> pair_bucket, pair_window;
        std::vector<uint8_t> pair_rc;
        std::vector<segment_info_t> pair_seg;
        for (size_t i = 0; i < sequence_ids_orig.size(); i++) {
            if (i >= buckets_.size()) break;   // padding bucket ids (NB > kept buckets) hold no sequence
            auto push = [&](const segment_info_t &id, bool rc) {
                const uint32_t w = window_of(id);
                if (!window_has_samples[w]) return;
                pair_bucket.push_back(static_cast<uint32_t>(i));
                pair_window.push_back(w);
                pair_rc.push_back(rc ? 1 : 0);
                pair_seg.push_back(id);
            };
            for (auto &id : sequence_ids_orig[i]) push(id, false);
            auto &rev = sequence_ids_rev_comp[i];
            for (auto it = rev.rbegin(); it != rev.rend(); ++it) push(*it, true);
        }
        const float prep_s = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        std::vector<int32_t> offsets(pair_bucket.size());
        std::vector<uint32_t> votes(pair_bucket.size());
        if (!pair_bucket.empty())
            _s->scan(sample_hash.data(), sample_pos.data(), segment_length.data(), static_cast<uint32_t>(segment_length.size()),
                     pair_bucket.data(), pair_window.data(), pair_rc.data(), static_cast<uint32_t>(pair_bucket.size()),
                     offsets.data(), votes.data());
        const float scan_s = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();

        std::vector<std::vector<locate_t>> res(_m->num_records);
        for (size_t i = 0; i < pair_bucket.size(); i++) {
            const int offset = offsets[i];
            if (offset <= 0) continue;                                                      // :674,686
            const segment_info_t &id = pair_seg[i];
            if (!pair_rc[i]) {
                res[id.first].push_back(std::make_tuple(pair_bucket[i], offset - id.second,
                                                        static_cast<unsigned int>(id.second), votes[i], true));
            } else {
                // get_read_length / get_segment_length return uint16_t in the reference (:53-60)
                const int segment_offset_ = static_cast<uint16_t>(read_lengths[id.first]) - id.second -
                                            static_cast<uint16_t>(segment_length[pair_window[i]]);
                res[id.first].push_back(std::make_tuple(pair_bucket[i], offset - segment_offset_,
                                                        static_cast<unsigned int>(id.second), votes[i], false));
            }
        }
        // the reference times the per-bucket index build and the offset search separately; both are
        // one device pass here, reported under the second label
        std::cerr << "[BENCHMARK]\tTotal time used for building k-mer index for each bucket: " << prep_s << " s.\n";
        std::cerr << "[BENCHMARK]\tTotal time used for finding exact location of the sequences: " << scan_s << " s ("
                  << scan_s * 1000 * 1000 / _m->num_records << " μs/seq).\n";
        return res;
    }

    // .bucket_id -> @SQ lines and per-bucket offsets (:473-503).  One line per kept bucket; the reference name of
    // a bucket is its line up to the first blank, consecutive buckets of one name form one @SQ entry whose
    // length is the upper bound #buckets * bucket_len (:491,502) -- two FASTA records whose headers agree up to
    // the first blank therefore share an entry and the second one's offsets carry on from the first's.
    struct sam_header {
        std::vector<std::string> bucket_name, ref_ids;
        std::vector<unsigned int> bucket_offsets;
        std::vector<size_t> ref_lengths;
    };
    sam_header read_bucket_ids(std::filesystem::path const &index_file) const {
        sam_header h;
        std::ifstream in(index_file);
        std::string line;
        for (size_t b = 0; b < buckets_.size(); b++) {
            line.clear();
            std::getline(in, line);
            line.resize(std::min(line.size(), line.find(' ')));
            h.bucket_name.push_back(line);
        }
        // runs of equal consecutive names: one entry per run, offsets counted from the run's first bucket
        for (size_t b = 0, e = 0; b < h.bucket_name.size(); b = e) {
            for (e = b; e < h.bucket_name.size() && h.bucket_name[e] == h.bucket_name[b]; e++)
                h.bucket_offsets.push_back(static_cast<unsigned int>(e - b) * bucket_length);
            h.ref_ids.push_back(h.bucket_name[b]);
            h.ref_lengths.push_back((e - b) * static_cast<size_t>(bucket_length));
        }
        return h;
    }

    // locate (:455-611): one SAM record per surviving location.  Without a verifier this is the plain
    // `bucketmap` branch (best locations by votes, MAPQ from votes, no CIGAR); with one it is the BM_ALIGN
    // branch (every location aligned, MAPQ = 60 + score, dropped below `quality_threshold`, CIGAR written).
    void locate(const std::string &sequence_file, std::filesystem::path const &index_file,
                std::filesystem::path const &sam_file, unsigned int quality_threshold = 30) {
        auto locate_res = locate_reads(sequence_file);
        const sam_header h = read_bucket_ids(index_file);

        sam_text sam(sam_file);
        sam.header(h.ref_ids, h.ref_lengths);
        unsigned int read_id = 0, mapped_locations = 0;
        auto t0 = std::chrono::steady_clock::now();
        if (_v) {
            mapped_locations = write_verified(sequence_file, locate_res, h, sam, quality_threshold, read_id);
        } else if (std::shared_ptr<FastqFile> fq = FastqFile::open(sequence_file)) {
            // the records of a block of reads are formatted by a thread of their own (blocks of 16 384 reads, rounds of up to
            // eight) and appended in read order: formatting 640 MB of SAM text on one thread was the tool's longest stage
            const size_t n = fq->count(), block = 16384;
            const unsigned T = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
            std::vector<std::string> text(T);
            std::vector<unsigned int> made(T);
            auto format_block = [&](unsigned t, size_t r0, size_t r1) {
                std::string &out = text[t], seq;
                out.clear();
                made[t] = 0;
                for (size_t r = r0; r < r1; r++) {
                    const FastqRecord rec = fq->view(r);
                    auto best = filter_best_locations(locate_res[r], static_cast<unsigned int>(rec.seq.size()));   // :540
                    if (best.empty()) continue;
                    seq.clear();
                    append_dna4(seq, rec.seq);
                    for (auto &[bucket_id, offset, segment_offset, votes, is_original] : best) {
                        (void)segment_offset;
                        const unsigned int map_qual = std::min(60u, 6 * votes);                      // :591
                        const size_t ref_offset = static_cast<size_t>(h.bucket_offsets[bucket_id]) + offset;  // :592, 0-based
                        sam_text::format(out, rec.id, is_original ? 0 : 16, h.bucket_name[bucket_id], ref_offset + 1, map_qual, "*", seq, rec.qual);
                        made[t]++;
                    }
                }
            };
            for (size_t r0 = 0; r0 < n; r0 += T * block) {
                std::vector<std::thread> pool;
                unsigned used = 0;
                for (unsigned t = 0; t < T && r0 + t * block < n; t++, used++)
                    if (t) pool.emplace_back(format_block, t, r0 + t * block, std::min(n, r0 + (t + 1) * block));
                format_block(0, r0, std::min(n, r0 + block));
                for (auto &th : pool) th.join();
                for (unsigned t = 0; t < used; t++) {
                    sam.append(text[t]);
                    mapped_locations += made[t];
                }
            }
            read_id = static_cast<unsigned int>(n);
        } else {
            std::string seq;
            for_each_fastq_stream(sequence_file, [&](const FastqRecord &rec) {
                auto best = filter_best_locations(locate_res[read_id], static_cast<unsigned int>(rec.seq.size()));   // :540
                if (!best.empty()) {
                    seq.clear();
                    append_dna4(seq, rec.seq);
                }
                for (auto &[bucket_id, offset, segment_offset, votes, is_original] : best) {
                    (void)segment_offset;
                    const unsigned int map_qual = std::min(60u, 6 * votes);                      // :591
                    const size_t ref_offset = static_cast<size_t>(h.bucket_offsets[bucket_id]) + offset;  // :592, 0-based
                    sam.record(rec.id, is_original ? 0 : 16, h.bucket_name[bucket_id], ref_offset + 1, map_qual, "*", seq, rec.qual);
                    mapped_locations++;
                }
                read_id++;
            });
        }
        sam.close();                                             // (every record is in the file before the time is taken)
        const float s = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        std::cerr << "[BENCHMARK]\tTotal mapped locations: " << mapped_locations << " ("
                  << static_cast<float>(mapped_locations) / read_id << " per sequence).\n";
        std::cerr << "[BENCHMARK]\tTotal time used for alignment verification and output: " << s << " s ("
                  << s / mapped_locations * 1000 * 1000 << " μs per pairwise alignment).\n";
    }

private:
    // BM_ALIGN branch of locate (:520-528,544-589), a block of reads at a time: the text window of every
    // location (:549-550), one batch of alignments, then the records in read order.
    //   * no _filter_best_locations in this branch (:538-541);
    //   * width = min(len + 1 + (size_t)(n * len), bucket size - offset), the product in float32 (:550);
    //   * map_qual = 60u + score in UNSIGNED arithmetic: below -60 it wraps to ~2^32 and so passes the
    //     `< quality_threshold` test (:570-573); the SAM field is 8 bits wide, hence the truncation;
    //   * POS = begin + bucket offset + offset + 1, also on the reverse strand, where `begin` counts in the
    //     reverse-complemented window (:576) -- kept as the reference computes it.
    // Deviation: a negative `offset` indexes before the bucket in the reference (undefined behaviour); here
    // the window is clipped to start at the bucket's first base.
    unsigned int write_verified(const std::string &sequence_file, const std::vector<std::vector<locate_t>> &locate_res,
                                const sam_header &h, sam_text &sam, unsigned int quality_threshold, unsigned int &read_id) {
        // Two blocks of reads: while the verifier works on one (a thread of its own), the records of the one before are
        // written and the next is read from the file.
        struct held { std::string id, seq, qual; uint64_t start; };
        struct Block {
            std::vector<held> reads;
            std::vector<uint8_t> bases;
            std::vector<uint64_t> text_start, query_start, cigar_offset;
            std::vector<uint32_t> text_len, query_len, begin, cigar;
            std::vector<uint8_t> text_rc;
            std::vector<int32_t> score;
            unsigned int first_read = 0;
            std::exception_ptr failed;
        } blocks[2];
        unsigned int mapped_locations = 0;
        size_t block_reads = 1u << 17;
        const size_t block_bases = 64u << 20;
        if (const char *e = std::getenv("BM_VERIFY_BLOCK_READS")) block_reads = std::max<size_t>(1, std::strtoull(e, nullptr, 10));   // tests
        int cur = 0, in_flight = -1, to_write = -1;
        blocks[0].first_read = read_id;
        std::thread worker;
        auto align = [&](Block &b) {                            // on the worker thread
            try {
                if (!b.text_start.empty())
                    _v->align(b.bases.data(), b.bases.size(), b.text_start.data(), b.text_len.data(), b.text_rc.data(),
                              b.query_start.data(), b.query_len.data(), static_cast<uint32_t>(b.text_start.size()), b.score, b.begin,
                              b.cigar_offset, b.cigar);
            } catch (...) {
                b.failed = std::current_exception();
            }
        };
        // BM_DUMP_ALIGNMENTS=<file> (tests): every alignment the verifier returned, kept or not, one line each --
        // read, text start in the concatenated genome, text length, strand, query length, score, begin, CIGAR
        std::ofstream dump;
        if (const char *e = std::getenv("BM_DUMP_ALIGNMENTS")) dump.open(e);
        auto write = [&](Block &b) {
            if (b.failed) std::rethrow_exception(b.failed);
            size_t a = 0;
            std::string cg;
            for (size_t r = 0; r < b.reads.size(); r++) {
                for (auto &[bucket_id, offset, segment_offset, votes, is_original] : locate_res[b.first_read + r]) {
                    (void)segment_offset; (void)votes;
                    if (dump.is_open()) {
                        dump << b.first_read + r << ' ' << b.text_start[a] << ' ' << b.text_len[a] << ' ' << int(b.text_rc[a]) << ' '
                             << b.query_len[a] << ' ' << b.score[a] << ' ' << b.begin[a] << ' ';
                        for (uint64_t x = b.cigar_offset[a]; x < b.cigar_offset[a + 1]; x++) dump << (b.cigar[x] >> 4) << "MID"[b.cigar[x] & 15u];
                        dump << (b.cigar_offset[a] == b.cigar_offset[a + 1] ? "*\n" : "\n");
                    }
                    const unsigned int wrapped = 60u + static_cast<unsigned int>(b.score[a]);        // :570
                    const size_t map_qual = wrapped;
                    if (!(map_qual < quality_threshold)) {                                            // :571-573
                        const int clipped = offset < 0 ? 0 : offset;
                        const size_t ref_offset = static_cast<size_t>(b.begin[a]) + h.bucket_offsets[bucket_id] + clipped;   // :576
                        cg.clear();
                        for (uint64_t x = b.cigar_offset[a]; x < b.cigar_offset[a + 1]; x++) {
                            cg += std::to_string(b.cigar[x] >> 4);
                            cg += "MID"[b.cigar[x] & 15u];
                        }
                        if (cg.empty()) cg = "*";
                        sam.record(b.reads[r].id, is_original ? 0 : 16, h.bucket_name[bucket_id], ref_offset + 1,
                                   static_cast<uint8_t>(map_qual), cg, b.reads[r].seq, b.reads[r].qual);
                        mapped_locations++;
                    }
                    a++;
                }
            }
            b.reads.clear(); b.bases.clear();
            b.text_start.clear(); b.text_len.clear(); b.text_rc.clear(); b.query_start.clear(); b.query_len.clear();
        };
        // the block just filled goes to the verifier as soon as the one before has left it; that one's records are written
        // while the verifier works
        auto flush = [&]() {
            if (in_flight >= 0) {
                worker.join();
                to_write = in_flight;
            }
            in_flight = cur;
            worker = std::thread(align, std::ref(blocks[cur]));
            cur ^= 1;
            if (to_write >= 0) {
                write(blocks[to_write]);
                to_write = -1;
            }
            blocks[cur].first_read = read_id;
        };
        try {
            for_each_fastq(sequence_file, [&](const FastqRecord &rec) {
                Block &b = blocks[cur];
                const size_t len = rec.seq.size();
                b.reads.push_back({std::string(rec.id), std::string(), std::string(rec.qual), b.bases.size()});
                append_dna4(b.reads.back().seq, rec.seq);
                b.bases.insert(b.bases.end(), rec.seq.begin(), rec.seq.end());
                for (auto &[bucket_id, offset, segment_offset, votes, is_original] : locate_res[read_id]) {
                    (void)segment_offset; (void)votes;
                    const size_t clipped = offset < 0 ? 0 : static_cast<size_t>(offset);
                    const size_t bucket_size = blen_[bucket_id];
                    const size_t width = std::min(len + 1 + static_cast<size_t>(allowed_indel_rate * len),   // :550
                                                  bucket_size - std::min(clipped, bucket_size));
                    b.text_start.push_back(bstart_[bucket_id] + std::min(clipped, bucket_size));
                    b.text_len.push_back(static_cast<uint32_t>(width));
                    b.text_rc.push_back(is_original ? 0 : 1);                                               // :563-567
                    b.query_start.push_back(b.reads.back().start);
                    b.query_len.push_back(static_cast<uint32_t>(len));
                }
                read_id++;
                if (b.reads.size() >= block_reads || b.bases.size() >= block_bases) flush();
            });
            flush();                                            // the last, possibly empty, block
            worker.join();
            in_flight = -1;
            write(blocks[cur ^ 1]);
        } catch (...) {
            if (worker.joinable()) worker.join();
            throw;
        }
        return mapped_locations;
    }
};

}  // namespace bm
