// gpu_offset_scanner.h -- the MI355X locator scan (include/bml.h) behind bm::offset_scanner.
// Replaces _prepare_read_query's sampling (bucket_map/locator/bucket_locator.h:317-343) and
// _create_kmer_index + _find_offset (:162-177,209-290) for all candidates of one _locate pass.  Fails loudly (throws) when the device path fails: no CPU fallback.
#pragma once

#include "../../include/bml.h"
#include "bucket_locator.h"

namespace bm {

class gpu_offset_scanner : public offset_scanner {
    bml_ctx *ctx_ = nullptr;

public:
    gpu_offset_scanner(uint32_t k, uint32_t num_samples, int allowed_mismatch, int allowed_indel,
                       uint32_t max_bucket_bases, int device = 0) {
        bml_params p{};
        p.k = k;
        p.num_samples = num_samples;
        p.allowed_mismatch = allowed_mismatch;
        p.allowed_indel = allowed_indel;
        p.max_bucket_bases = max_bucket_bases;
        p.device = device;
        if (bml_create(&p, &ctx_) != BML_OK)
            throw std::runtime_error(std::string("cannot create the GPU locator scan: ") + bml_last_error());
    }
    ~gpu_offset_scanner() override { bml_destroy(ctx_); }

    void load_genome(const uint8_t *bases, uint64_t n_bases, const uint64_t *bucket_start, const uint32_t *bucket_len,
                     uint32_t n_buckets) override {
        if (bml_load_genome(ctx_, bases, n_bases, bucket_start, bucket_len, n_buckets) != BML_OK)
            throw std::runtime_error(std::string("uploading the genome failed: ") + bml_last_error());
    }

    void sample_windows(const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes, const uint64_t *win_start,
                        const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality, uint32_t *out_hash,
                        uint16_t *out_pos, uint8_t *out_has) override {
        if (bml_sample_windows(ctx_, bases, quals, n_bytes, win_start, win_len, n_windows, min_base_quality, out_hash,
                               out_pos, out_has) != BML_OK)
            throw std::runtime_error(std::string("the GPU k-mer sampling failed: ") + bml_last_error());
    }

    void scan(const uint32_t *sample_hash, const uint16_t *sample_pos, const uint32_t *seg_len, uint32_t n_windows,
              const uint32_t *pair_bucket, const uint32_t *pair_window, const uint8_t *pair_rc, uint32_t n_pairs,
              int32_t *out_offset, uint32_t *out_votes) override {
        if (bml_locate(ctx_, sample_hash, sample_pos, seg_len, n_windows, pair_bucket, pair_window, pair_rc, n_pairs,
                       out_offset, out_votes) != BML_OK)
            throw std::runtime_error(std::string("the GPU locator scan failed: ") + bml_last_error());
        float a = 0, b = 0, c = 0;
        uint64_t n = 0;
        bml_last_stats(ctx_, &a, &b, &c, &n);
        std::cerr << "[BENCHMARK]\tGPU locator scan: " << n_pairs << " candidates, " << n << " k-mer occurrences; scan " << a
                  << " ms, sort " << b << " ms, vote replay " << c << " ms.\n";
    }
};

}  // namespace bm
