// gpu_q_gram_mapper.h -- q_gram_mapper<NB> re-hosted on the MI355X filter (libbmf.so).
//
// Same constructor arguments, same load / map / reset, same stderr lines as the reference's
// q_gram_mapper (bucket_map/mapper/q_gram_mapper.h:204-646), except that NB is a run-time value.
// map() keeps the reference's host work -- FASTQ loop, long-read windowing, scatter into per-bucket
// lists (q_gram_mapper.h:483-557) -- and hands the arithmetic of query_sequence (:414-480) to
// bmf_map_windows in batches.  With several devices the windows of a batch are split into contiguous
// ranges, one host thread + context per device (index replicated, no collective), and merged in order.
#pragma once

#include "../../include/bmf.h"
#include "bm_genome.h"
#include "mapper.h"

#include <chrono>
#include <iostream>
#include <thread>

namespace bm {

// The host half of q_gram_mapper::map, independent of where query_sequence runs.
class batched_mapper : public mapper {
protected:
    unsigned int num_buckets_, read_length_, num_segment_samples_, max_candidates_;
    size_t batch_reads_ = 1u << 20;

    // query_sequence for n windows (views into bases/quals); counts: 2 per window, buckets:
    // 2 x max_candidates per window.  Returns false on failure (message already printed).
    virtual bool query_windows(const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                               const uint64_t *win_start, const uint32_t *win_len, uint32_t n, uint32_t *counts,
                               uint32_t *buckets) = 0;
    virtual bool index_loaded() const = 0;

public:
    batched_mapper(unsigned int num_buckets, unsigned int read_len, unsigned int num_candidate_buckets,
                   unsigned int num_segment_samples)
        : num_buckets_(num_buckets), read_length_(read_len), num_segment_samples_(num_segment_samples),
          max_candidates_(num_candidate_buckets) {}

    // q_gram_mapper::map (q_gram_mapper.h:483-557)
    std::pair<segments_t, segments_t> map(std::filesystem::path const &sequence_file) override {
        unsigned int mapped_reads = 0, num_buckets_orig = 0, num_buckets_rev_comp = 0;
        segments_t res_orig(num_buckets_), res_rev_comp(num_buckets_);
        const unsigned int first_record = num_records;
        auto t0 = std::chrono::steady_clock::now();

        std::vector<uint8_t> bases, quals;
        std::vector<uint64_t> win_start;
        std::vector<uint32_t> win_len, win_read;
        std::vector<int> win_pos;
        std::vector<uint32_t> counts, buckets, starts(num_segment_samples_ ? num_segment_samples_ : 1);
        std::vector<uint8_t> read_mapped;
        size_t reads_in_batch = 0;

        auto flush = [&]() {
            const uint32_t n = static_cast<uint32_t>(win_start.size());
            if (n == 0) return;
            counts.assign(2 * static_cast<size_t>(n), 0);
            buckets.assign(2 * static_cast<size_t>(n) * max_candidates_, 0);
            if (!index_loaded()) {
                // q_gram_mapper.h:389-393 (printed once per query in the reference; once per batch here)
                std::cerr << "[ERROR]\t\tThe q-gram index is empty. Cannot accept query.\n";
            } else if (!query_windows(bases.data(), quals.data(), bases.size(), win_start.data(), win_len.data(), n,
                                      counts.data(), buckets.data())) {
                throw std::runtime_error("the candidate-bucket filter failed (see the [ERROR] line above)");
            }
            read_mapped.assign(reads_in_batch, 0);
            const unsigned int base_read = win_read.empty() ? 0 : win_read.front();
            for (uint32_t w = 0; w < n; w++) {
                const segment_info_t seg{win_read[w], win_pos[w]};
                const uint32_t cf = counts[2 * w], cr = counts[2 * w + 1];
                const uint32_t *bf = buckets.data() + static_cast<size_t>(2 * w) * max_candidates_;
                const uint32_t *br = bf + max_candidates_;
                for (uint32_t i = 0; i < cf; i++) res_orig[bf[i]].push_back(seg);
                for (uint32_t i = 0; i < cr; i++) res_rev_comp[br[i]].push_back(seg);
                if (cf || cr) {
                    read_mapped[win_read[w] - base_read] = 1;
                    num_buckets_orig += cf;
                    num_buckets_rev_comp += cr;
                }
            }
            for (uint8_t m : read_mapped) mapped_reads += m;
            bases.clear(); quals.clear(); win_start.clear(); win_len.clear(); win_read.clear(); win_pos.clear();
            reads_in_batch = 0;
        };

        for_each_fastq(sequence_file.string(), [&](const FastqRecord &rec) {
            const uint64_t off = bases.size();
            const uint32_t len = static_cast<uint32_t>(rec.seq.size());
            bases.insert(bases.end(), rec.seq.begin(), rec.seq.end());
            quals.insert(quals.end(), rec.qual.begin(), rec.qual.end());
            // q_gram_mapper.h:510-523: window starts {0}, or Sampler(5) for reads longer than 2*read_len
            const uint32_t nw = bmf_window_starts(len, read_length_, num_segment_samples_, starts.data());
            for (uint32_t i = 0; i < nw; i++) {
                const uint32_t s = starts[i];
                win_start.push_back(off + s);
                win_len.push_back(std::min(s + read_length_, len) - s);
                win_read.push_back(num_records);
                win_pos.push_back(static_cast<int>(s));
            }
            ++num_records;
            if (++reads_in_batch >= batch_reads_) flush();
        });
        flush();

        const float time = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0f;
        const unsigned int n_rec = num_records - first_record;
        // q_gram_mapper.h:548-555 (the reference divides by the running num_records)
        std::cerr << "[BENCHMARK]\tElapsed time for bucket mapping: " << time << " s (" << time * 1000 * 1000 / num_records
                  << " μs/seq).\n";
        std::cerr << "[BENCHMARK]\tNumber of reads that have at least one candidate bucket: " << mapped_reads << "  ("
                  << static_cast<float>(mapped_reads) / num_records * 100 << "%).\n";
        std::cerr << "[BENCHMARK]\tAverage number of buckets an original read is mapped to: "
                  << static_cast<float>(num_buckets_orig) / mapped_reads << ".\n";
        std::cerr << "[BENCHMARK]\tAverage number of buckets a reverse complement of the read is mapped to: "
                  << static_cast<float>(num_buckets_rev_comp) / mapped_reads << ".\n";
        (void)n_rec;
        return std::make_pair(std::move(res_orig), std::move(res_rev_comp));
    }
};

class gpu_q_gram_mapper : public batched_mapper {
    std::vector<bmf_ctx *> ctx_;
    bool loaded_ = false;

protected:
    bool index_loaded() const override { return loaded_; }

    bool query_windows(const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes, const uint64_t *win_start,
                       const uint32_t *win_len, uint32_t n, uint32_t *counts, uint32_t *buckets) override {
        const size_t D = ctx_.size();
        std::vector<int> rc(D, BMF_OK);
        std::vector<std::string> msg(D);
        auto work = [&](size_t d) {
            const uint32_t w0 = static_cast<uint32_t>(static_cast<uint64_t>(n) * d / D);
            const uint32_t w1 = static_cast<uint32_t>(static_cast<uint64_t>(n) * (d + 1) / D);
            rc[d] = bmf_map_windows(ctx_[d], bases, quals, n_bytes, win_start + w0, win_len + w0, w1 - w0,
                                    counts + 2 * static_cast<size_t>(w0),
                                    buckets + 2 * static_cast<size_t>(w0) * max_candidates_);
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
