// make_mapper_oracle.cpp -- TEST INFRASTRUCTURE: the CPU oracles behind bm::mapper and bm::offset_scanner.
//
// Links oracle/bm_oracle.c and oracle/bm_locator_oracle.c into the same `bucketmap` main as the product,
// so that (a) the host plumbing (FASTQ loop, windowing, scatter, result ordering, SAM) can be exercised
// without a GPU and (b) the SAM file of the GPU build can be compared with the SAM file of an
// oracle-backed build on the same inputs.  Never shipped: built only into tests/cpp/bucketmap_oracle.
#include "../../bucket-map_amd/host/bm_indexer.h"
#include "../../bucket-map_amd/host/bucket_locator.h"
#include "../../bucket-map_amd/host/cli.h"
#include "../../bucket-map_amd/host/gpu_q_gram_mapper.h"   // for bm::batched_mapper (host half of map())
#include "../../oracle/bm_align_oracle.h"
#include "../../oracle/bm_locator_oracle.h"
#include "../../oracle/bm_oracle.h"

#include <memory>
#include <thread>

namespace {

class oracle_mapper : public bm::batched_mapper {
    bmo_params p_{};
    bmo_index *ix_ = nullptr;

protected:
    bool index_loaded() const override { return ix_ != nullptr; }
    bool query_windows(const uint8_t *text, uint64_t, const uint64_t *seq_start, const uint64_t *qual_start,
                       const uint32_t *win_len, uint32_t n, uint32_t *counts, std::vector<uint32_t> &ids) override {
        std::vector<uint32_t> dense(2 * static_cast<size_t>(n) * p_.max_candidates);
        std::vector<uint8_t> bases, quals;
        std::vector<uint64_t> win_start;
        gather_windows(text, seq_start, qual_start, win_len, n, bases, quals, win_start);
        bmo_map_windows(ix_, bases.data(), quals.data(), win_start.data(), win_len, n, counts, dense.data());
        for (size_t i = 0; i < 2 * static_cast<size_t>(n); i++)
            ids.insert(ids.end(), dense.begin() + i * p_.max_candidates, dense.begin() + i * p_.max_candidates + counts[i]);
        return true;
    }

public:
    oracle_mapper(const bm::cmd_arguments &a, unsigned int num_buckets, unsigned int fault)
        : bm::batched_mapper(num_buckets, a.max_read_length, 30, 5) {
        p_.num_buckets = num_buckets;
        p_.q = a.index_seed_length;
        p_.k = a.query_seed_length;
        p_.num_samples = a.mapper_sample_size;
        p_.num_fault = fault;
        p_.threshold = bmo_threshold(a.mapper_distinguishability_threshold, num_buckets);
        p_.min_base_quality = a.average_base_quality * a.query_seed_length;
        p_.max_candidates = 30;
        p_.read_len = a.max_read_length;
        p_.num_segment_samples = 5;
    }
    ~oracle_mapper() override { bmo_index_destroy(ix_); }
    void load(std::filesystem::path const &dir, const std::string &indicator) override {
        if (ix_) return;
        ix_ = bmo_index_load(&p_, dir.string().c_str(), indicator.c_str());
    }
    void reset() override {
        bmo_index_destroy(ix_);
        ix_ = nullptr;
    }
};

class oracle_scanner : public bm::offset_scanner {
    bmlo_params p_{};
    std::vector<uint8_t> genome_;
    std::vector<uint64_t> bstart_;
    std::vector<uint32_t> blen_;

public:
    oracle_scanner(uint32_t k, uint32_t num_samples, int allowed_mismatch, int allowed_indel) {
        p_.k = k;
        p_.num_samples = num_samples;
        p_.allowed_mismatch = allowed_mismatch;
        p_.allowed_indel = allowed_indel;
    }
    void load_genome(const uint8_t *bases, uint64_t n_bases, const uint64_t *bucket_start, const uint32_t *bucket_len,
                     uint32_t n_buckets) override {
        genome_.assign(bases, bases + n_bases);
        bstart_.assign(bucket_start, bucket_start + n_buckets);
        blen_.assign(bucket_len, bucket_len + n_buckets);
    }
    void sample_windows(const uint8_t *bases, const uint8_t *quals, uint64_t, const uint64_t *win_start,
                        const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality, uint32_t *out_hash,
                        uint16_t *out_pos, uint8_t *out_has) override {
        bmlo_sample_windows(p_.k, p_.num_samples, min_base_quality, bases, quals, win_start, win_len, n_windows, out_hash,
                            out_pos, out_has);
    }
    void scan(const uint32_t *sample_hash, const uint16_t *sample_pos, const uint32_t *seg_len, uint32_t,
              const uint32_t *pair_bucket, const uint32_t *pair_window, const uint8_t *pair_rc, uint32_t n_pairs,
              int32_t *out_offset, uint32_t *out_votes) override {
        if (bmlo_locate(&p_, genome_.data(), bstart_.data(), blen_.data(), static_cast<uint32_t>(blen_.size()), sample_hash,
                        sample_pos, seg_len, pair_bucket, pair_window, pair_rc, n_pairs, out_offset, out_votes))
            throw std::runtime_error("oracle locator: bad bucket id");
    }
};

class oracle_verifier : public bm::alignment_verifier {
    std::vector<uint8_t> genome_;

public:
    void load_genome(const uint8_t *bases, uint64_t n_bases) override { genome_.assign(bases, bases + n_bases); }
    void align(const uint8_t *reads, uint64_t, const uint64_t *text_start, const uint32_t *text_len, const uint8_t *text_rc,
               const uint64_t *query_start, const uint32_t *query_len, uint32_t n, std::vector<int32_t> &score,
               std::vector<uint32_t> &begin, std::vector<uint64_t> &cigar_offset, std::vector<uint32_t> &cigar) override {
        score.assign(n, 0);
        begin.assign(n, 0);
        cigar_offset.assign(static_cast<size_t>(n) + 1, 0);
        uint64_t cap = 0;
        for (uint32_t a = 0; a < n; a++) cap += static_cast<uint64_t>(text_len[a]) + query_len[a] + 1;
        cigar.assign(cap, 0);
        if (bmao_align_batch(genome_.data(), reads, text_start, text_len, text_rc, query_start, query_len, n, score.data(),
                             begin.data(), cigar_offset.data(), cigar.data(), cap))
            throw std::runtime_error("oracle verifier failed");
        cigar.resize(cigar_offset[n]);
    }
};

}  // namespace

std::unique_ptr<bm::alignment_verifier> bm_make_verifier(const bm::cmd_arguments &) { return std::make_unique<oracle_verifier>(); }

std::unique_ptr<bm::mapper> bm_make_mapper(const bm::cmd_arguments &args, unsigned int num_buckets, unsigned int fault) {
    return std::make_unique<oracle_mapper>(args, num_buckets, fault);
}

// the oracle-backed tool always uses the host indexer
bool bm_gpu_index(const bm::cmd_arguments &, const bm::Genome &, unsigned int, bm::QgramIndex &) { return false; }
std::thread bm_warm_up(const bm::cmd_arguments &) { return {}; }
void bm_report_resources(const bm::cmd_arguments &) {}

std::unique_ptr<bm::offset_scanner> bm_make_scanner(const bm::cmd_arguments &args, int allowed_mismatch, int allowed_indel) {
    return std::make_unique<oracle_scanner>(args.query_seed_length, static_cast<uint32_t>(args.locator_sample_size),
                                            allowed_mismatch, allowed_indel);
}
