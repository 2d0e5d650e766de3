// gpu_alignment_verifier.h -- the MI355X alignment verifier (include/bmv.h) behind bm::alignment_verifier.
// Replaces seqan3::align_pairwise in the BM_ALIGN branch of bucket_locator::locate
// (bucket_map/locator/bucket_locator.h:520-528,569-576) for a whole block of candidates per call.
// Several devices: the genome is replicated and the block's alignments -- independent of each other, the
// reference runs them one by one (:560-589) -- are cut into contiguous ranges of equal cell count, one per
// device; each device is handed only the span of the read buffer its queries cover.
// Fails loudly (throws) when the device path fails: no CPU fallback.
#pragma once

#include "../../include/bmv.h"
#include "bucket_locator.h"
#include "device_pool.h"

namespace bm {

class gpu_alignment_verifier : public alignment_verifier {
    std::vector<bmv_ctx *> ctx_;

    static void check(int rc, const char *what) {
        if (rc != BMV_OK) throw std::runtime_error(std::string(what) + bmv_last_error());
    }

public:
    explicit gpu_alignment_verifier(std::vector<int> devices = {0}) {
        bmv_params p{};
        p.max_query_len = 65536;   // the ABI's limits: reads are not known yet
        p.max_text_len = 81920;
        for (int dev : devices) {
            p.device = dev;
            bmv_ctx *c = nullptr;
            if (bmv_create(&p, &c) != BMV_OK) {
                const std::string why = bmv_last_error();
                for (bmv_ctx *o : ctx_) bmv_destroy(o);
                throw std::runtime_error("cannot create the GPU alignment verifier on device " + std::to_string(dev) + ": " + why);
            }
            ctx_.push_back(c);
        }
    }
    ~gpu_alignment_verifier() override {
        for (bmv_ctx *c : ctx_) bmv_destroy(c);
    }

    void load_genome_records(const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records) override {
        for_each_device(ctx_.size(), [&](size_t d) { check(bmv_load_genome_records(ctx_[d], rec, rec_len, n_records), "uploading the genome failed: "); });
    }
    void load_genome(const uint8_t *bases, uint64_t n_bases) override {
        for_each_device(ctx_.size(), [&](size_t d) { check(bmv_load_genome(ctx_[d], bases, n_bases), "uploading the genome failed: "); });
    }

    void align(const uint8_t *reads, uint64_t n_read_bytes, const uint64_t *text_start, const uint32_t *text_len,
               const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len, uint32_t n,
               std::vector<int32_t> &score, std::vector<uint32_t> &begin, std::vector<uint64_t> &cigar_offset,
               std::vector<uint32_t> &cigar) override {
        const size_t D = ctx_.size();
        const auto t0 = std::chrono::steady_clock::now();
        score.assign(n, 0);
        begin.assign(n, 0);
        cigar_offset.assign(static_cast<size_t>(n) + 1, 0);
        const std::vector<uint32_t> cut =
            cut_by_cost(n, D, [&](uint32_t a) { return static_cast<uint64_t>(query_len[a]) * text_len[a] + 1u; });
        std::vector<uint64_t> total(D, 0), cells(D, 0);
        std::vector<float> ms(D, 0.f);
        for_each_device(D, [&](size_t d) {
            const uint32_t a0 = cut[d], m = cut[d + 1] - cut[d];
            if (m == 0) return;
            if (D == 1) {
                check(bmv_align(ctx_[0], reads, n_read_bytes, text_start, text_len, text_rc, query_start, query_len, n, &total[0]),
                      "the GPU alignment verifier failed: ");
            } else {
                uint64_t lo = ~0ull, hi = 0;
                for (uint32_t a = a0; a < a0 + m; a++) {
                    lo = std::min(lo, query_start[a]);
                    hi = std::max(hi, query_start[a] + query_len[a]);
                }
                std::vector<uint64_t> rebased(query_start + a0, query_start + a0 + m);
                for (uint64_t &s : rebased) s -= lo;
                check(bmv_align(ctx_[d], reads + lo, hi - lo, text_start + a0, text_len + a0, text_rc + a0, rebased.data(),
                                query_len + a0, m, &total[d]), "the GPU alignment verifier failed: ");
            }
            bmv_last_stats(ctx_[d], &ms[d], &cells[d]);
        });
        // CIGARs of the ranges back to back, in range order; a range's offsets count from its own first entry
        std::vector<uint64_t> at(D + 1, 0);
        for (size_t d = 0; d < D; d++) at[d + 1] = at[d] + total[d];
        cigar.assign(at[D], 0);
        for_each_device(D, [&](size_t d) {
            const uint32_t a0 = cut[d], m = cut[d + 1] - cut[d];
            if (m == 0) return;
            std::vector<uint64_t> off(static_cast<size_t>(m) + 1);
            check(bmv_results(ctx_[d], score.data() + a0, begin.data() + a0, off.data(), cigar.data() + at[d]),
                  "reading the verifier's results failed: ");
            for (uint32_t a = 0; a < m; a++) cigar_offset[a0 + a] = at[d] + off[a];
        });
        cigar_offset[n] = at[D];
        float slowest = 0;
        uint64_t all_cells = 0;
        for (size_t d = 0; d < D; d++) {
            slowest = std::max(slowest, ms[d]);
            all_cells += cells[d];
        }
        const float call_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::cerr << "[BENCHMARK]\tGPU alignment verification: " << n << " alignments, " << all_cells << " cells; kernels " << slowest
                  << " ms" << (D > 1 ? " on the slowest of " + std::to_string(D) + " devices," : "") << " of " << call_ms
                  << " ms in the call.\n";
    }
};

}  // namespace bm
