// gpu_q_gram_mapper.h -- the reference-side binding: the file a BucketMap maintainer drops into
// bucket_map/mapper/ to put the MI355X candidate-bucket filter behind the reference's own `mapper` interface.
//
// It is written against the REFERENCE's declarations, not this repository's copies: it expects `class mapper`
// (bucket_map/mapper/mapper.h:4-34) and `segment_info_t` / `segments_t` (bucket_map/utils.h:309-311) to be declared
// before it is included, exactly as bucket_map/mapper/q_gram_mapper.h gets them (`#include "./mapper.h"`,
// `#include "../utils.h"`), and it needs nothing else from the reference tree -- no SeqAn3: the FASTQ loop below is
// plain C++ (four-line records; a maintainer may swap `for_each_record` for `seqan3::sequence_file_input`, the
// `rec.sequence() | seqan3::views::to_char` of q_gram_mapper.h:506-509).  Link with -lbmf (this repository's
// bucket-map_amd/libbmf.so, declared in include/bmf.h).
//
// tests/test_integration_binding.py compiles it against the real /root/reference/bucket_map/mapper/mapper.h and uses
// it through a `mapper*` the way bucket_locator does (bucket_locator.h:449,624,627,642).
//
// The standalone tool of this repository uses bucket-map_amd/host/gpu_q_gram_mapper.h instead: the same calls plus
// double-buffered batches, several devices and the reference's benchmark-only entry points.
#ifndef BUCKET_MAP_GPU_Q_GRAM_MAPPER_H
#define BUCKET_MAP_GPU_Q_GRAM_MAPPER_H

#include <bmf.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

class gpu_q_gram_mapper : public mapper {
    bmf_ctx *ctx = nullptr;
    unsigned int num_buckets, read_length, num_segment_samples, max_candidates;
    bool loaded = false;
    std::size_t batch_reads = 1u << 18, batch_bytes = 64u << 20;

    // one batch of reads: bases and qualities back to back, one entry per window
    struct batch_t {
        std::vector<uint8_t> bases, quals;
        std::vector<uint64_t> win_start;
        std::vector<uint32_t> win_len;
        std::vector<segment_info_t> win_info;   // (read id, window start inside the read), q_gram_mapper.h:526
        std::size_t n_reads = 0;
        void clear() { bases.clear(); quals.clear(); win_start.clear(); win_len.clear(); win_info.clear(); n_reads = 0; }
    };

    template <typename F>
    static void for_each_record(std::filesystem::path const &fastq, F &&f) {
        std::ifstream is(fastq);
        if (!is) throw std::runtime_error("cannot open " + fastq.string());
        std::string id, seq, plus, qual;
        while (std::getline(is, id)) {
            if (id.empty()) continue;
            if (id[0] != '@' || !std::getline(is, seq) || !std::getline(is, plus) || !std::getline(is, qual) ||
                plus.empty() || plus[0] != '+' || qual.size() != seq.size())
                throw std::runtime_error("malformed FASTQ record near " + id);
            f(seq, qual);
        }
    }

    // query_sequence (q_gram_mapper.h:414-480) for every window of the batch, then the scatter of :526-538
    void flush(batch_t &b, segments_t &res_orig, segments_t &res_rev_comp, unsigned int &mapped_reads,
               unsigned int &num_buckets_orig, unsigned int &num_buckets_rev_comp, std::vector<uint32_t> &counts,
               std::vector<uint32_t> &ids) {
        const uint32_t n = static_cast<uint32_t>(b.win_start.size());
        if (n == 0) return;
        if (!loaded) {
            std::cerr << "[ERROR]\t\tThe q-gram index is empty. Cannot accept query.\n";   // q_gram_mapper.h:389-393
            b.clear();
            return;
        }
        counts.assign(2 * static_cast<std::size_t>(n), 0);
        ids.resize(2 * static_cast<std::size_t>(n) * max_candidates);
        uint64_t used = 0;
        if (bmf_map_windows_compact(ctx, b.bases.data(), b.quals.data(), b.bases.size(), b.win_start.data(), b.win_len.data(),
                                    n, counts.data(), ids.data(), ids.size(), &used) != BMF_OK)
            throw std::runtime_error(std::string("bmf_map_windows_compact: ") + bmf_last_error());
        const uint32_t *next = ids.data();
        unsigned int last_mapped = ~0u;
        for (uint32_t w = 0; w < n; w++) {
            const uint32_t cf = counts[2 * w], cr = counts[2 * w + 1];
            for (uint32_t i = 0; i < cf; i++) res_orig[*next++].push_back(b.win_info[w]);
            for (uint32_t i = 0; i < cr; i++) res_rev_comp[*next++].push_back(b.win_info[w]);
            if (cf + cr) {
                if (b.win_info[w].first != last_mapped) ++mapped_reads;
                last_mapped = b.win_info[w].first;
                num_buckets_orig += cf;
                num_buckets_rev_comp += cr;
            }
        }
        b.clear();
    }

public:
    // the arguments of q_gram_mapper's constructor (q_gram_mapper.h:281-289) with NB first (a template argument there)
    gpu_q_gram_mapper(unsigned int bucket_num, unsigned int bucket_len, unsigned int read_len, uint8_t k_, uint8_t q_,
                      unsigned int samples, unsigned int fault, float distinguishability,
                      unsigned int quality_threshold = 35, unsigned int num_candidate_buckets = 30,
                      unsigned int num_segment_samples_ = 5, int device = 0, unsigned int flags = BMF_FLAG_EARLY_EXIT)
        : mapper(), num_buckets(bucket_num), read_length(read_len), num_segment_samples(num_segment_samples_),
          max_candidates(num_candidate_buckets) {
        (void)bucket_len;
        bmf_params p{};
        p.num_buckets = bucket_num;
        p.q = q_;
        p.k = k_;
        p.num_samples = samples;
        p.num_fault = fault;
        p.threshold = bmf_threshold(distinguishability, bucket_num);   // q_gram_mapper.h:163
        p.min_base_quality = quality_threshold * k_;                   // q_gram_mapper.h:303
        p.max_candidates = num_candidate_buckets;
        p.read_len = read_len;
        p.num_segment_samples = num_segment_samples_;
        p.device = device;
        p.flags = flags;
        if (bmf_create(&p, &ctx) != BMF_OK) throw std::runtime_error(std::string("bmf_create: ") + bmf_last_error());
    }
    gpu_q_gram_mapper(const gpu_q_gram_mapper &) = delete;
    gpu_q_gram_mapper &operator=(const gpu_q_gram_mapper &) = delete;
    ~gpu_q_gram_mapper() { bmf_destroy(ctx); }

    // q_gram_mapper::load (q_gram_mapper.h:318-372)
    void load(std::filesystem::path const &index_directory, const std::string &indicator) {
        if (loaded) {
            std::cerr << "[ERROR]\t\tThe q-gram index is not empty. Terminating load.\n";
            return;
        }
        const int rc = bmf_load_index_files(ctx, index_directory.string().c_str(), indicator.c_str());
        if (rc == BMF_ERR_IO) return;   // a missing file leaves the index empty there too (:332-333,348-349)
        if (rc != BMF_OK) throw std::runtime_error(std::string("bmf_load_index_files: ") + bmf_last_error());
        loaded = true;
        std::cerr << "[INFO]\t\tSuccessfully loaded " << index_directory / (indicator + ".qgram") << ".\n";
    }

    // q_gram_mapper::map (q_gram_mapper.h:483-557)
    std::pair<segments_t, segments_t> map(std::filesystem::path const &sequence_file) {
        segments_t res_orig(num_buckets), res_rev_comp(num_buckets);
        unsigned int mapped_reads = 0, num_buckets_orig = 0, num_buckets_rev_comp = 0;
        const auto t0 = std::chrono::steady_clock::now();
        batch_t b;
        std::vector<uint32_t> counts, ids, starts(num_segment_samples ? num_segment_samples : 1);
        for_each_record(sequence_file, [&](const std::string &seq, const std::string &qual) {
            const uint32_t len = static_cast<uint32_t>(seq.size());
            if (b.n_reads && (b.n_reads >= batch_reads || b.bases.size() + len > batch_bytes))
                flush(b, res_orig, res_rev_comp, mapped_reads, num_buckets_orig, num_buckets_rev_comp, counts, ids);
            const uint64_t base = b.bases.size();
            b.bases.insert(b.bases.end(), seq.begin(), seq.end());
            b.quals.insert(b.quals.end(), qual.begin(), qual.end());
            // :510-523: the whole read (truncated to read_length), or Sampler(5) window starts for reads > 2 * read_length
            const uint32_t nw = bmf_window_starts(len, read_length, num_segment_samples, starts.data());
            for (uint32_t i = 0; i < nw; i++) {
                b.win_start.push_back(base + starts[i]);
                b.win_len.push_back(std::min(starts[i] + read_length, len) - starts[i]);
                b.win_info.push_back(segment_info_t(num_records, static_cast<int>(starts[i])));
            }
            ++b.n_reads;
            ++num_records;
        });
        flush(b, res_orig, res_rev_comp, mapped_reads, num_buckets_orig, num_buckets_rev_comp, counts, ids);
        const float time = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0f;
        std::cerr << "[BENCHMARK]\tElapsed time for bucket mapping: " << time << " s (" << time * 1000 * 1000 / num_records << " μs/seq).\n";
        std::cerr << "[BENCHMARK]\tNumber of reads that have at least one candidate bucket: " << mapped_reads << "  ("
                  << ((float)mapped_reads) / num_records * 100 << "%).\n";
        std::cerr << "[BENCHMARK]\tAverage number of buckets an original read is mapped to: " << ((float)num_buckets_orig) / mapped_reads << ".\n";
        std::cerr << "[BENCHMARK]\tAverage number of buckets a reverse complement of the read is mapped to: "
                  << ((float)num_buckets_rev_comp) / mapped_reads << ".\n";
        return std::make_pair(std::move(res_orig), std::move(res_rev_comp));
    }

    // q_gram_mapper::reset (q_gram_mapper.h:638-645)
    void reset() {
        bmf_reset(ctx);
        loaded = false;
    }
};

#endif
