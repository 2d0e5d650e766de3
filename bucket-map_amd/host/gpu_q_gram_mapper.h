// gpu_q_gram_mapper.h -- q_gram_mapper<NB> re-hosted on the MI355X filter (libbmf.so).
//
// Same constructor arguments, same load / map / reset, same stderr lines as the reference's
// q_gram_mapper (bucket_map/mapper/q_gram_mapper.h:204-646), except that NB is a run-time value.
// map() keeps the reference's host work -- FASTQ loop, long-read windowing, scatter into per-bucket
// lists (q_gram_mapper.h:483-557) -- and hands the arithmetic of query_sequence (:414-480) to
// bmf_map_text_windows_compact in batches (the windows are views into the mapped FASTQ file).  With several devices the windows of a batch are split into contiguous
// ranges, one host thread + context per device (index replicated, no collective), and merged in order.
#pragma once

#include "../../include/bmf.h"
#include "bm_genome.h"
#include "mapper.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <thread>

namespace bm {

// The host half of q_gram_mapper::map, independent of where query_sequence runs.
//
// The FASTQ file is mapped and indexed ONCE (FastqFile, shared with the locator's passes) and nothing of it is copied on
// this side: a batch is the window views of a range of reads -- (offset of the bases, offset of the qualities, length) into
// the mapped file -- and query_windows gathers them where it needs them (the GPU library: into page-locked piece buffers,
// by a few threads, while the piece before is on the device).  Two batch slots: while the devices work on one batch, the
// windows of the next are laid out.  The per-bucket lists are built at the END, by a stable counting sort over all
// batches' results (threads own bucket ranges), so they keep the reference's (read, window) order (q_gram_mapper.h:526-533)
// and no list is ever grown entry by entry.
class batched_mapper : public mapper {
protected:
    unsigned int num_buckets_, read_length_, num_segment_samples_, max_candidates_;
    size_t batch_reads_ = 1u << 18;   // reads per batch (BM_BATCH_READS) -- the first two are an eighth and a half of it, so that the
                                      // devices start early; a call's fixed cost (filling and draining its pipeline) is a piece's worth
    size_t batch_bytes_ = 96u << 20;  // ... and bases per batch (BM_BATCH_MB): 260 000 reads of 300 bp, 9 600 of 10 kbp
    bool ramp_ = true;                // (off when BM_BATCH_READS is given: tests count on exact batch sizes)
    bool log_batches_ = std::getenv("BM_LOG_BATCHES") != nullptr;   // one stderr line per batch: when it went out, how long it took

    // query_sequence for n windows: window w = text[seq_start[w], +win_len[w]) with qualities text[qual_start[w], +win_len[w]);
    // counts: 2 per window (read as-is, reverse complement), ids: the candidate lists back to back in that order (resized
    // by the callee).  Returns false on failure (message already printed).
    virtual bool query_windows(const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start, const uint64_t *qual_start,
                               const uint32_t *win_len, uint32_t n, uint32_t *counts, std::vector<uint32_t> &ids) = 0;
    virtual bool index_loaded() const = 0;

    // for implementations that want the windows back to back (bases and qualities at the same offsets)
    static void gather_windows(const uint8_t *text, const uint64_t *seq_start, const uint64_t *qual_start, const uint32_t *win_len,
                               uint32_t n, std::vector<uint8_t> &bases, std::vector<uint8_t> &quals, std::vector<uint64_t> &start) {
        start.resize(n);
        uint64_t at = 0;
        for (uint32_t w = 0; w < n; w++) {
            start[w] = at;
            at += win_len[w];
        }
        bases.resize(at + 1);
        quals.resize(at + 1);
        for (uint32_t w = 0; w < n; w++) {
            std::memcpy(bases.data() + start[w], text + seq_start[w], win_len[w]);
            std::memcpy(quals.data() + start[w], text + qual_start[w], win_len[w]);
        }
    }

private:
    struct Slot {
        std::vector<uint8_t> own;            // pipes only: the batch's sequences and qualities, copied from the stream
        std::vector<uint64_t> seq_start, qual_start;
        std::vector<uint32_t> win_len, win_read, counts, ids;
        std::vector<int> win_pos;
        size_t n_reads = 0, n_bases = 0;
        bool ok = true;
        void clear() {
            own.clear(); seq_start.clear(); qual_start.clear(); win_len.clear(); win_read.clear(); win_pos.clear();
            n_reads = n_bases = 0;
        }
    };
    // what a finished batch leaves behind for the final scatter
    struct Done {
        std::vector<uint32_t> win_read, counts, ids;
        std::vector<int> win_pos;
    };

public:
    batched_mapper(unsigned int num_buckets, unsigned int read_len, unsigned int num_candidate_buckets,
                   unsigned int num_segment_samples)
        : num_buckets_(num_buckets), read_length_(read_len), num_segment_samples_(num_segment_samples),
          max_candidates_(num_candidate_buckets) {
        if (const char *e = std::getenv("BM_BATCH_READS")) {
            batch_reads_ = std::max<size_t>(1, std::strtoull(e, nullptr, 10));
            ramp_ = false;
        }
        if (const char *e = std::getenv("BM_BATCH_MB")) batch_bytes_ = std::max<size_t>(1, std::strtoull(e, nullptr, 10)) << 20;
    }

    // q_gram_mapper::map (q_gram_mapper.h:483-557)
    std::pair<segments_t, segments_t> map(std::filesystem::path const &sequence_file) override {
        auto t0 = std::chrono::steady_clock::now();
        std::shared_ptr<FastqFile> fq = FastqFile::open(sequence_file.string());   // nullptr: a pipe
        const uint8_t *text = fq ? reinterpret_cast<const uint8_t *>(fq->data()) : nullptr;
        const uint64_t text_bytes = fq ? fq->size() : 0;
        Slot slots[2];
        std::vector<Done> done;
        std::thread worker;
        int cur = 0, in_flight = -1;
        size_t n_submitted = 0;
        // the batch being filled is full at: an eighth of a batch, half a batch, then whole batches
        auto reads_limit = [&]() { return !ramp_ || n_submitted >= 2 ? batch_reads_ : std::max<size_t>(1, batch_reads_ >> (n_submitted ? 1 : 3)); };
        auto bytes_limit = [&]() { return !ramp_ || n_submitted >= 2 ? batch_bytes_ : std::max<size_t>(1, batch_bytes_ >> (n_submitted ? 1 : 3)); };
        std::vector<uint32_t> starts(num_segment_samples_ ? num_segment_samples_ : 1);

        // runs on the worker thread: query_sequence for every window of the slot
        // (an exception must not leave a std::thread: it is turned into a failed batch, reported by collect)
        auto run = [&](Slot &s) {
            s.ok = false;
            try {
                const uint32_t n = static_cast<uint32_t>(s.seq_start.size());
                s.counts.assign(2 * static_cast<size_t>(n), 0);
                s.ids.clear();
                s.ok = true;
                if (n == 0) return;
                if (!index_loaded()) {
                    // q_gram_mapper.h:389-393 (printed once per query in the reference; once per batch here)
                    std::cerr << "[ERROR]\t\tThe q-gram index is empty. Cannot accept query.\n";
                } else {
                    const auto q0 = std::chrono::steady_clock::now();
                    const uint8_t *src = fq ? text : s.own.data();
                    s.ok = query_windows(src, fq ? text_bytes : s.own.size(), s.seq_start.data(), s.qual_start.data(), s.win_len.data(), n,
                                         s.counts.data(), s.ids);
                    if (log_batches_)
                        std::cerr << "[bm] batch of " << s.n_reads << " reads, " << n << " windows, " << s.n_bases << " bases: filter "
                                  << std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - q0).count()
                                  << " ms, " << std::chrono::duration<float, std::milli>(q0 - t0).count() << " ms after map() began\n";
                }
            } catch (const std::exception &e) {
                std::cerr << "[ERROR]\t\t" << e.what() << "\n";
                s.ok = false;
            }
        };
        // keeps a finished batch's results for the final scatter, in batch order
        // ... and counts what it adds to every list (this runs beside the next batch's device work)
        std::vector<uint32_t> n_orig(num_buckets_, 0), n_rev(num_buckets_, 0);
        unsigned int mapped_reads = 0, num_buckets_orig = 0, num_buckets_rev_comp = 0, last_mapped = ~0u;
        auto collect = [&](Slot &s) {
            if (!s.ok) throw std::runtime_error("the candidate-bucket filter failed (see the [ERROR] line above)");
            const size_t n = s.win_read.size();
            const uint32_t *next = s.ids.data();
            for (size_t w = 0; w < n; w++) {
                const uint32_t cf = s.counts[2 * w], cr = s.counts[2 * w + 1];
                if (cf || cr) {
                    if (s.win_read[w] != last_mapped) ++mapped_reads;
                    last_mapped = s.win_read[w];
                    num_buckets_orig += cf;
                    num_buckets_rev_comp += cr;
                    for (uint32_t i = 0; i < cf; i++) ++n_orig[next[i]];
                    for (uint32_t i = cf; i < cf + cr; i++) ++n_rev[next[i]];
                    next += cf + cr;
                }
            }
            done.emplace_back();
            Done &d = done.back();
            d.win_read.swap(s.win_read);
            d.win_pos.swap(s.win_pos);
            d.counts.swap(s.counts);
            d.ids.swap(s.ids);
            s.clear();
        };
        auto finish_in_flight = [&]() {
            if (in_flight < 0) return;
            worker.join();
            collect(slots[in_flight]);
            in_flight = -1;
        };
        auto submit = [&]() {   // hand the filled slot to the devices, continue laying out windows in the other one
            finish_in_flight();
            in_flight = cur;
            worker = std::thread(run, std::ref(slots[cur]));
            cur ^= 1;
            ++n_submitted;
        };
        // q_gram_mapper.h:510-523: window starts {0}, or Sampler(5) for reads longer than 2*read_len
        auto add_read = [&](uint64_t seq_off, uint64_t qual_off, uint32_t len) {
            Slot &s = slots[cur];
            if (len <= 2 * read_length_) {
                s.seq_start.push_back(seq_off);
                s.qual_start.push_back(qual_off);
                s.win_len.push_back(std::min(read_length_, len));
                s.win_read.push_back(num_records);
                s.win_pos.push_back(0);
            } else {
                const uint32_t nw = bmf_window_starts(len, read_length_, num_segment_samples_, starts.data());
                for (uint32_t i = 0; i < nw; i++) {
                    const uint32_t st = starts[i];
                    s.seq_start.push_back(seq_off + st);
                    s.qual_start.push_back(qual_off + st);
                    s.win_len.push_back(std::min(st + read_length_, len) - st);
                    s.win_read.push_back(num_records);
                    s.win_pos.push_back(static_cast<int>(st));
                }
            }
            s.n_bases += len;
            ++num_records;
            if (++s.n_reads >= reads_limit()) submit();
        };

        try {
            if (fq) {
                for (size_t i = 0;;) {
                    const size_t n = fq->wait(i + 1);
                    if (n <= i) break;
                    for (; i < n; i++) {
                        const FastqFile::Rec &r = fq->rec(i);
                        if (slots[cur].n_reads && slots[cur].n_bases + r.len > bytes_limit()) submit();   // the batch is full by bytes
                        add_read(r.seq, r.qual, r.len);
                    }
                }
            } else {
                for_each_fastq_stream(sequence_file.string(), [&](const FastqRecord &rec) {
                    const uint32_t len = static_cast<uint32_t>(rec.seq.size());
                    if (slots[cur].n_reads && slots[cur].n_bases + len > bytes_limit()) submit();
                    Slot &s = slots[cur];
                    const uint64_t at = s.own.size();
                    s.own.insert(s.own.end(), rec.seq.begin(), rec.seq.end());
                    s.own.insert(s.own.end(), rec.qual.begin(), rec.qual.end());
                    add_read(at, at + len, len);
                });
            }
            if (slots[cur].n_reads) submit();
            finish_in_flight();
        } catch (...) {
            if (worker.joinable()) worker.join();
            throw;
        }

        // q_gram_mapper.h:526-538: (read, window start) into the per-bucket lists, in (batch, window) order = read order.
        // A stable counting sort: threads own bucket ranges, count their lists' lengths, size them once, fill them.
        const auto t_batches = std::chrono::steady_clock::now();
        segments_t res_orig(num_buckets_), res_rev_comp(num_buckets_);
        {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const unsigned T = static_cast<unsigned>(std::min<size_t>(std::min(16u, hw), std::max<size_t>(1, (static_cast<size_t>(num_buckets_orig) + num_buckets_rev_comp) / 50000)));
            auto scatter = [&](unsigned t) {
                const uint32_t b0 = static_cast<uint32_t>(static_cast<uint64_t>(num_buckets_) * t / T);
                const uint32_t b1 = static_cast<uint32_t>(static_cast<uint64_t>(num_buckets_) * (t + 1) / T);
                for (uint32_t b = b0; b < b1; b++) {                 // every list sized once
                    if (n_orig[b]) res_orig[b].reserve(n_orig[b]);
                    if (n_rev[b]) res_rev_comp[b].reserve(n_rev[b]);
                }
                for (const Done &d : done) {
                    const size_t n = d.win_read.size();
                    const uint32_t *next = d.ids.data();
                    for (size_t w = 0; w < n; w++) {
                        const uint32_t cf = d.counts[2 * w], cr = d.counts[2 * w + 1];
                        for (uint32_t i = 0; i < cf + cr; i++) {
                            const uint32_t b = next[i];
                            if (b >= b0 && b < b1) (i < cf ? res_orig : res_rev_comp)[b].emplace_back(d.win_read[w], d.win_pos[w]);
                        }
                        next += cf + cr;
                    }
                }
            };
            if (T <= 1) {
                scatter(0);
            } else {
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < T; t++) pool.emplace_back(scatter, t);
                scatter(0);
                for (auto &th : pool) th.join();
            }
        }

        if (log_batches_)
            std::cerr << "[bm] all batches done " << std::chrono::duration<float, std::milli>(t_batches - t0).count()
                      << " ms after map() began; per-bucket lists built in "
                      << std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_batches).count() << " ms\n";
        const float time = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        // q_gram_mapper.h:548-555 (the reference divides by the running num_records)
        std::cerr << "[BENCHMARK]\tElapsed time for bucket mapping: " << time << " s (" << time * 1000 * 1000 / num_records
                  << " μs/seq).\n";
        std::cerr << "[BENCHMARK]\tNumber of reads that have at least one candidate bucket: " << mapped_reads << "  ("
                  << static_cast<float>(mapped_reads) / num_records * 100 << "%).\n";
        std::cerr << "[BENCHMARK]\tAverage number of buckets an original read is mapped to: "
                  << static_cast<float>(num_buckets_orig) / mapped_reads << ".\n";
        std::cerr << "[BENCHMARK]\tAverage number of buckets a reverse complement of the read is mapped to: "
                  << static_cast<float>(num_buckets_rev_comp) / mapped_reads << ".\n";
        return std::make_pair(std::move(res_orig), std::move(res_rev_comp));
    }

    // The reference's two benchmark-only entry points (used by bucket_map/mapper_test.cpp:61-63).
    using query_result_t = std::vector<std::pair<std::vector<unsigned int>, std::vector<unsigned int>>>;

    // q_gram_mapper::_query_file (q_gram_mapper.h:560-578): query_sequence on every whole record -- no windowing, no
    // truncation, so records longer than the read length the mapper was built for are an error here.
    query_result_t _query_file(std::filesystem::path const &sequence_file) {
        query_result_t res;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<uint8_t> bases, quals;
        std::vector<uint64_t> win_start, qual_start;
        std::vector<uint32_t> win_len, counts, ids;
        auto flush = [&]() {
            const uint32_t n = static_cast<uint32_t>(win_start.size());
            if (n == 0) return;
            counts.assign(2 * static_cast<size_t>(n), 0);
            ids.clear();
            // one text: the batch's sequences, then its qualities
            const uint64_t n_bases = bases.size();
            bases.insert(bases.end(), quals.begin(), quals.end());
            qual_start.resize(n);
            for (uint32_t w = 0; w < n; w++) qual_start[w] = n_bases + win_start[w];
            if (!index_loaded()) {
                std::cerr << "[ERROR]\t\tThe q-gram index is empty. Cannot accept query.\n";
            } else if (!query_windows(bases.data(), bases.size(), win_start.data(), qual_start.data(), win_len.data(), n, counts.data(), ids)) {
                throw std::runtime_error("the candidate-bucket filter failed (see the [ERROR] line above)");
            }
            const uint32_t *next = ids.data();
            for (uint32_t w = 0; w < n; w++) {
                const uint32_t *bf = next, *br = next + counts[2 * w];
                next = br + counts[2 * w + 1];
                res.emplace_back(std::vector<unsigned int>(bf, br), std::vector<unsigned int>(br, next));
            }
            bases.clear(); quals.clear(); win_start.clear(); win_len.clear();
        };
        for_each_fastq(sequence_file.string(), [&](const FastqRecord &rec) {
            win_start.push_back(bases.size());
            win_len.push_back(static_cast<uint32_t>(rec.seq.size()));
            bases.insert(bases.end(), rec.seq.begin(), rec.seq.end());
            quals.insert(quals.end(), rec.qual.begin(), rec.qual.end());
            if (win_start.size() >= batch_reads_) flush();
        });
        flush();
        const float time = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0f;
        std::cerr << "[BENCHMARK]\tElapsed time for bucket query: " << time << " s (" << time * 1000 * 1000 / res.size() << " μs/seq).\n";
        return res;
    }

    // q_gram_mapper::_check_ground_truth (q_gram_mapper.h:580-636): `bucket offset is_rev_comp cigar` per read, as the
    // read simulator writes its .bucket_ground_truth file.
    void _check_ground_truth(const query_result_t &query_results, std::filesystem::path const &ground_truth_file) const {
        std::ifstream is(ground_truth_file);
        int bucket = 0, exact_location = 0, correct_map = 0, total_bucket_numbers = 0;
        std::string cigar;
        bool rev_comp = false;
        std::vector<int> with_n_buckets(2 * static_cast<size_t>(max_candidates_) + 1, 0);
        for (const auto &[buckets_orig, buckets_rev_comp] : query_results) {
            is >> bucket >> exact_location >> rev_comp >> cigar;
            const auto &own = rev_comp ? buckets_rev_comp : buckets_orig;
            if (std::find(own.begin(), own.end(), static_cast<unsigned int>(bucket)) != own.end()) correct_map++;
            total_bucket_numbers += static_cast<int>(buckets_orig.size() + buckets_rev_comp.size());
            ++with_n_buckets[buckets_orig.size() + buckets_rev_comp.size()];
        }
        const float n = static_cast<float>(query_results.size());
        auto line = [&](const char *what, int v) {
            std::cerr << "[BENCHMARK]\t" << what << v << " (" << v / n * 100 << "%).\n";
        };
        auto up_to = [&](size_t m) {
            int s = 0;
            for (size_t i = 1; i <= m && i < with_n_buckets.size(); i++) s += with_n_buckets[i];
            return s;
        };
        std::cerr << "[BENCHMARK]\tTotal number of sequences: " << query_results.size() << ".\n";
        line("Correct bucket predictions: ", correct_map);
        std::cerr << "[BENCHMARK]\tAverage number of buckets returned: " << total_bucket_numbers / n << ".\n";
        line("Number of sequences that have no candidate bucket: ", with_n_buckets[0]);
        line("Number of uniquely mapped sequences: ", with_n_buckets.size() > 1 ? with_n_buckets[1] : 0);
        line("Number of sequences mapped to <= 5 buckets: ", up_to(5));
        line("Number of sequences mapped to <= 10 buckets: ", up_to(10));
    }
};

class gpu_q_gram_mapper : public batched_mapper {
    std::vector<bmf_ctx *> ctx_;
    std::vector<std::unique_ptr<uint32_t[]>> ids_buf_;   // per device: the packed candidate lists of its window range
    std::vector<uint64_t> ids_cap_;
    bool loaded_ = false;
    float distinguishability_ = 0.5f;

protected:
    bool index_loaded() const override { return loaded_; }
    bool query_windows(const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start, const uint64_t *qual_start,
                       const uint32_t *win_len, uint32_t n, uint32_t *counts, std::vector<uint32_t> &ids) override {
        const size_t D = ctx_.size();
        std::vector<int> rc(D, BMF_OK);
        std::vector<std::string> msg(D);
        std::vector<uint64_t> used(D, 0);
        // every context is handed the shared text and ITS window range: bmf_map_text_windows_compact gathers and uploads only
        // those windows, and packs the range's candidate lists into the device's own buffer
        auto work = [&](size_t d) {
            const uint32_t w0 = static_cast<uint32_t>(static_cast<uint64_t>(n) * d / D);
            const uint32_t w1 = static_cast<uint32_t>(static_cast<uint64_t>(n) * (d + 1) / D);
            const uint64_t cap = 2ull * (w1 - w0) * max_candidates_;
            if (cap > ids_cap_[d]) {   // uninitialised on purpose: a worst case of 240 bytes per window, mostly never touched
                ids_buf_[d].reset(new uint32_t[cap]);
                ids_cap_[d] = cap;
            }
            rc[d] = bmf_map_text_windows_compact(ctx_[d], text, n_bytes, seq_start + w0, qual_start + w0, win_len + w0, w1 - w0,
                                                 counts + 2 * static_cast<size_t>(w0), ids_buf_[d].get(), cap, &used[d]);
            if (rc[d] != BMF_OK) msg[d] = bmf_last_error();
        };
        if (D == 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (size_t d = 0; d < D; d++) pool.emplace_back(work, d);
            for (auto &t : pool) t.join();
        }
        for (size_t d = 0; d < D; d++)
            if (rc[d] != BMF_OK) {
                std::cerr << "[ERROR]\t\tGPU " << d << ": " << msg[d] << "\n";
                return false;
            }
        for (size_t d = 0; d < D; d++) ids.insert(ids.end(), ids_buf_[d].get(), ids_buf_[d].get() + used[d]);   // contiguous ranges: device order is window order
        return true;
    }

public:
    // q_gram_mapper ctor (q_gram_mapper.h:281-308) + NB and the device list as run-time values.
    // Throws std::runtime_error if a device/context cannot be created: there is no CPU fallback.
    gpu_q_gram_mapper(unsigned int num_buckets, unsigned int bucket_len, unsigned int read_len,
                      uint8_t query_seed_length, uint8_t index_seed_length, unsigned int samples, unsigned int fault,
                      float distinguishability, unsigned int quality_threshold = 35,
                      unsigned int num_candidate_buckets = 30, unsigned int num_segment_samples = 5,
                      std::vector<int> devices = {0}, unsigned int flags = 0)
        : batched_mapper(num_buckets, read_len, num_candidate_buckets, num_segment_samples) {
        (void)bucket_len;
        distinguishability_ = distinguishability;
        std::cerr << "[INFO]\t\tSet query seed length to be " << static_cast<int>(query_seed_length)
                  << ", and index seed length " << static_cast<int>(index_seed_length) << ".\n";
        bmf_params p{};
        p.num_buckets = num_buckets;
        p.q = index_seed_length;
        p.k = query_seed_length;
        p.num_samples = samples;
        p.num_fault = fault;
        p.threshold = bmf_threshold(distinguishability, num_buckets);   // q_gram_mapper.h:163
        p.min_base_quality = quality_threshold * query_seed_length;    // q_gram_mapper.h:303
        p.max_candidates = num_candidate_buckets;
        p.read_len = read_len;
        p.num_segment_samples = num_segment_samples;
        p.flags = flags;
        for (int dev : devices) {
            p.device = dev;
            bmf_ctx *c = nullptr;
            if (bmf_create(&p, &c) != BMF_OK) {
                const std::string why = bmf_last_error();
                for (bmf_ctx *o : ctx_) bmf_destroy(o);
                throw std::runtime_error("cannot create the GPU filter on device " + std::to_string(dev) + ": " + why);
            }
            ctx_.push_back(c);
        }
        ids_buf_.resize(ctx_.size());
        ids_cap_.assign(ctx_.size(), 0);
    }

    ~gpu_q_gram_mapper() override {
        for (bmf_ctx *c : ctx_) bmf_destroy(c);
    }

    // q_gram_mapper::load (q_gram_mapper.h:318-372)
    void load(std::filesystem::path const &index_directory, const std::string &indicator) override {
        if (loaded_) {
            std::cerr << "[ERROR]\t\tThe q-gram index is not empty. Terminating load.\n";
            return;
        }
        auto t0 = std::chrono::steady_clock::now();
        for (bmf_ctx *c : ctx_) {
            const int rc = bmf_load_index_files(c, index_directory.string().c_str(), indicator.c_str());
            if (rc == BMF_ERR_IO) {
                // the reference silently leaves the index empty when a file is missing (:332-333,348-349)
                std::cerr << "[WARNING]\t" << bmf_last_error() << "\n";
                for (bmf_ctx *o : ctx_) bmf_reset(o);
                return;
            }
            if (rc != BMF_OK) throw std::runtime_error(std::string("loading the index failed: ") + bmf_last_error());
        }
        loaded_ = true;
        // the staging buffers of map()'s calls, now: its first batch then pays for neither page-locking nor device allocations
        for (bmf_ctx *c : ctx_) {
            const size_t per_call = std::min<size_t>(batch_reads_, batch_bytes_ / std::max(1u, read_length_)) / ctx_.size() + 1;
            if (bmf_map_reserve(c, static_cast<uint32_t>(std::min<size_t>(per_call, 1u << 24)), 1) != BMF_OK)
                std::cerr << "[WARNING]\t" << bmf_last_error() << "\n";
        }
        {   // distinguishability_filter::read's log line (q_gram_mapper.h:171-186): rows with MORE zeros than the threshold
            uint64_t n_rows = 0;
            if (bmf_index_download(ctx_[0], nullptr, &n_rows) == BMF_OK && n_rows) {
                std::vector<uint32_t> zeros(n_rows);
                if (bmf_index_zeros(ctx_[0], zeros.data()) == BMF_OK) {
                    const unsigned int threshold = bmf_threshold(distinguishability_, num_buckets_);
                    const size_t valid = static_cast<size_t>(std::count_if(zeros.begin(), zeros.end(), [&](uint32_t z) { return z > threshold; }));
                    std::cerr << "[BENCHMARK]\tNumber of Q-grams with distinguishability >= " << static_cast<float>(threshold) / num_buckets_
                              << ": " << valid << " (" << static_cast<float>(valid) / n_rows * 100 << "%).\n";
                }
            }
        }
        const float s = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0f;
        std::cerr << "[INFO]\t\tSuccessfully loaded " << (index_directory / (indicator + ".kmers_index")) << ".\n";
        std::cerr << "[BENCHMARK]\tElapsed time for loading index files: " << s << " s.\n";
        std::cerr << "[INFO]\t\tSuccessfully loaded " << (index_directory / (indicator + ".qgram")) << ".\n";
    }

    // q_gram_mapper::reset (q_gram_mapper.h:638-645)
    void reset() override {
        for (bmf_ctx *c : ctx_) bmf_reset(c);
        loaded_ = false;
    }
};

}  // namespace bm
