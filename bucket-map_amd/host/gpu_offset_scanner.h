// gpu_offset_scanner.h -- the MI355X locator scan (include/bml.h) behind bm::offset_scanner.
// Replaces _prepare_read_query's sampling (bucket_map/locator/bucket_locator.h:317-343) and
// _create_kmer_index + _find_offset (:162-177,209-290) for all candidates of one _locate pass.
// Several devices: the genome is replicated, the windows (sampling) and the candidates (scan) are cut into
// contiguous ranges, one per device -- the reference's per-bucket loop (:651-695) has no dependence between
// candidates.  Fails loudly (throws) when the device path fails: no CPU fallback.
#pragma once

#include "../../include/bml.h"
#include "bucket_locator.h"
#include "device_pool.h"

namespace bm {

class gpu_offset_scanner : public offset_scanner {
    std::vector<bml_ctx *> ctx_;
    uint32_t num_samples_;

    static void check(int rc, const char *what) {
        if (rc != BML_OK) throw std::runtime_error(std::string(what) + bml_last_error());
    }

public:
    gpu_offset_scanner(uint32_t k, uint32_t num_samples, int allowed_mismatch, int allowed_indel,
                       uint32_t max_bucket_bases, std::vector<int> devices = {0})
        : num_samples_(num_samples) {
        bml_params p{};
        p.k = k;
        p.num_samples = num_samples;
        p.allowed_mismatch = allowed_mismatch;
        p.allowed_indel = allowed_indel;
        p.max_bucket_bases = max_bucket_bases;
        for (int dev : devices) {
            p.device = dev;
            bml_ctx *c = nullptr;
            if (bml_create(&p, &c) != BML_OK) {
                const std::string why = bml_last_error();
                for (bml_ctx *o : ctx_) bml_destroy(o);
                throw std::runtime_error("cannot create the GPU locator scan on device " + std::to_string(dev) + ": " + why);
            }
            ctx_.push_back(c);
        }
    }
    ~gpu_offset_scanner() override {
        for (bml_ctx *c : ctx_) bml_destroy(c);
    }

    void load_genome(const uint8_t *bases, uint64_t n_bases, const uint64_t *bucket_start, const uint32_t *bucket_len,
                     uint32_t n_buckets) override {
        for_each_device(ctx_.size(), [&](size_t d) {
            check(bml_load_genome(ctx_[d], bases, n_bases, bucket_start, bucket_len, n_buckets), "uploading the genome failed: ");
        });
    }

    void load_genome_records(const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records, const uint64_t *bucket_start,
                             const uint32_t *bucket_len, uint32_t n_buckets) override {
        for_each_device(ctx_.size(), [&](size_t d) {
            check(bml_load_genome_records(ctx_[d], rec, rec_len, n_records, bucket_start, bucket_len, n_buckets), "uploading the genome failed: ");
        });
    }

    void sample_windows(const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes, const uint64_t *win_start,
                        const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality, uint32_t *out_hash,
                        uint16_t *out_pos, uint8_t *out_has) override {
        const size_t D = ctx_.size();
        if (D == 1) {
            check(bml_sample_windows(ctx_[0], bases, quals, n_bytes, win_start, win_len, n_windows, min_base_quality, out_hash,
                                     out_pos, out_has), "the GPU k-mer sampling failed: ");
            return;
        }
        // every device gets the byte span its windows cover, window starts rebased to it
        const size_t p = num_samples_;
        const std::vector<uint32_t> cut = cut_evenly(n_windows, D);
        for_each_device(D, [&](size_t d) {
            const uint32_t w0 = cut[d], n = cut[d + 1] - cut[d];
            if (n == 0) return;
            uint64_t lo = ~0ull, hi = 0;
            for (uint32_t w = w0; w < w0 + n; w++) {
                lo = std::min(lo, win_start[w]);
                hi = std::max(hi, win_start[w] + win_len[w]);
            }
            std::vector<uint64_t> rebased(win_start + w0, win_start + w0 + n);
            for (uint64_t &s : rebased) s -= lo;
            check(bml_sample_windows(ctx_[d], bases + lo, quals + lo, hi - lo, rebased.data(), win_len + w0, n, min_base_quality,
                                     out_hash + w0 * p, out_pos + w0 * p, out_has + w0),
                  "the GPU k-mer sampling failed: ");
        });
    }

    void sample_text_windows(const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start, const uint64_t *qual_start,
                             const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality, uint32_t *out_hash,
                             uint16_t *out_pos, uint8_t *out_has) override {
        // every device is handed the shared text and ITS window range: it gathers and uploads those windows only
        const size_t D = ctx_.size(), p = num_samples_;
        const std::vector<uint32_t> cut = cut_evenly(n_windows, D);
        for_each_device(D, [&](size_t d) {
            const uint32_t w0 = cut[d], n = cut[d + 1] - cut[d];
            if (n == 0) return;
            check(bml_sample_text_windows(ctx_[d], text, n_bytes, seq_start + w0, qual_start + w0, win_len + w0, n, min_base_quality,
                                          out_hash + w0 * p, out_pos + w0 * p, out_has + w0),
                  "the GPU k-mer sampling failed: ");
        });
    }

    void scan(const uint32_t *sample_hash, const uint16_t *sample_pos, const uint32_t *seg_len, uint32_t n_windows,
              const uint32_t *pair_bucket, const uint32_t *pair_window, const uint8_t *pair_rc, uint32_t n_pairs,
              int32_t *out_offset, uint32_t *out_votes) override {
        const size_t D = ctx_.size();
        // candidates arrive grouped by bucket; any contiguous cut keeps them grouped (a bucket whose run straddles
        // a cut is simply scanned by both devices), and a candidate's result does not depend on its neighbours
        const std::vector<uint32_t> cut = cut_evenly(n_pairs, D);
        std::vector<float> ms(3 * D, 0.f);
        std::vector<uint64_t> occ(D, 0);
        const size_t p = num_samples_;
        for_each_device(D, [&](size_t d) {
            const uint32_t i0 = cut[d], n = cut[d + 1] - cut[d];
            if (n == 0) return;
            if (D == 1) {
                check(bml_locate(ctx_[d], sample_hash, sample_pos, seg_len, n_windows, pair_bucket, pair_window, pair_rc, n,
                                 out_offset, out_votes), "the GPU locator scan failed: ");
            } else {
                // a device is handed the samples of the windows ITS candidates name, renumbered in order of first use
                // (candidates are grouped by bucket, so a range of them names windows from all over the batch)
                std::vector<uint32_t> local(n_windows, 0xFFFFFFFFu), used, pw(n);
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t w = pair_window[i0 + i];
                    if (w >= n_windows) throw std::runtime_error("the GPU locator scan failed: a candidate names window " + std::to_string(w));
                    if (local[w] == 0xFFFFFFFFu) {
                        local[w] = static_cast<uint32_t>(used.size());
                        used.push_back(w);
                    }
                    pw[i] = local[w];
                }
                std::vector<uint32_t> hash(used.size() * p), len(used.size());
                std::vector<uint16_t> pos(used.size() * p);
                for (size_t u = 0; u < used.size(); u++) {
                    std::copy(sample_hash + used[u] * p, sample_hash + (used[u] + 1) * p, hash.begin() + u * p);
                    std::copy(sample_pos + used[u] * p, sample_pos + (used[u] + 1) * p, pos.begin() + u * p);
                    len[u] = seg_len[used[u]];
                }
                check(bml_locate(ctx_[d], hash.data(), pos.data(), len.data(), static_cast<uint32_t>(used.size()), pair_bucket + i0,
                                 pw.data(), pair_rc + i0, n, out_offset + i0, out_votes + i0), "the GPU locator scan failed: ");
            }
            bml_last_stats(ctx_[d], &ms[3 * d], &ms[3 * d + 1], &ms[3 * d + 2], &occ[d]);
        });
        float a = 0, b = 0, c = 0;
        uint64_t n = 0;
        for (size_t d = 0; d < D; d++) {   // the devices run side by side: report the slowest
            a = std::max(a, ms[3 * d]);
            b = std::max(b, ms[3 * d + 1]);
            c = std::max(c, ms[3 * d + 2]);
            n += occ[d];
        }
        std::cerr << "[BENCHMARK]\tGPU locator scan: " << n_pairs << " candidates, " << n << " k-mer occurrences; scan " << a
                  << " ms, host between kernels " << b << " ms, vote replay " << c << " ms" << (D > 1 ? " (slowest of " + std::to_string(D) + " devices)" : "")
                  << ".\n";
    }
};

}  // namespace bm
