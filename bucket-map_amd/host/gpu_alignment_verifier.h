// gpu_alignment_verifier.h -- the MI355X alignment verifier (include/bmv.h) behind bm::alignment_verifier.
// Replaces seqan3::align_pairwise in the BM_ALIGN branch of bucket_locator::locate
// (bucket_map/locator/bucket_locator.h:520-528,569-576) for a whole block of candidates per call.
// Fails loudly (throws) when the device path fails: no CPU fallback.
#pragma once

#include "../../include/bmv.h"
#include "bucket_locator.h"

namespace bm {

class gpu_alignment_verifier : public alignment_verifier {
    bmv_ctx *ctx_ = nullptr;

public:
    explicit gpu_alignment_verifier(int device = 0) {
        bmv_params p{};
        p.max_query_len = 65536;   // the ABI's limits: reads are not known yet
        p.max_text_len = 81920;
        p.device = device;
        if (bmv_create(&p, &ctx_) != BMV_OK)
            throw std::runtime_error(std::string("cannot create the GPU alignment verifier: ") + bmv_last_error());
    }
    ~gpu_alignment_verifier() override { bmv_destroy(ctx_); }

    void load_genome(const uint8_t *bases, uint64_t n_bases) override {
        if (bmv_load_genome(ctx_, bases, n_bases) != BMV_OK)
            throw std::runtime_error(std::string("uploading the genome failed: ") + bmv_last_error());
    }

    void align(const uint8_t *reads, uint64_t n_read_bytes, const uint64_t *text_start, const uint32_t *text_len,
               const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len, uint32_t n,
               std::vector<int32_t> &score, std::vector<uint32_t> &begin, std::vector<uint64_t> &cigar_offset,
               std::vector<uint32_t> &cigar) override {
        uint64_t total = 0;
        const auto t0 = std::chrono::steady_clock::now();
        if (bmv_align(ctx_, reads, n_read_bytes, text_start, text_len, text_rc, query_start, query_len, n, &total) != BMV_OK)
            throw std::runtime_error(std::string("the GPU alignment verifier failed: ") + bmv_last_error());
        score.assign(n, 0);
        begin.assign(n, 0);
        cigar_offset.assign(static_cast<size_t>(n) + 1, 0);
        cigar.assign(total, 0);
        if (bmv_results(ctx_, score.data(), begin.data(), cigar_offset.data(), cigar.data()) != BMV_OK)
            throw std::runtime_error(std::string("reading the verifier's results failed: ") + bmv_last_error());
        float ms = 0;
        uint64_t cells = 0;
        bmv_last_stats(ctx_, &ms, &cells);
        const float call_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::cerr << "[BENCHMARK]\tGPU alignment verification: " << n << " alignments, " << cells << " cells; kernels " << ms
                  << " ms of " << call_ms << " ms in the call.\n";
    }
};

}  // namespace bm
