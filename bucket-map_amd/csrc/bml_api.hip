// bml_api.hip -- C ABI (include/bml.h) over the locator-scan kernels in bml_kernels.hip.h.
// Host side only: chunking of candidates per bucket, buffers, the two replay kernels (one thread or one workgroup per
// candidate) and the automatic re-run when the occurrence buffer was too small.  No device-wide sort: the scan kernel
// writes every candidate's occurrences to a segment of their own, grouped by sample.
#include "bml_kernels.hip.h"
#include "bm_hip_util.h"

#include "../../include/bml.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(BML_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

void build_dna4_lut(uint8_t *lut) {
    memset(lut, 0, 256);
    const char *m[4] = {"AaRrWwMmDdHhVv", "CcYySsBb", "GgKk", "TtUu"};
    for (int r = 0; r < 4; r++)
        for (const char *c = m[r]; *c; c++) lut[(uint8_t)*c] = (uint8_t)r;
}

// Device buffer that only grows.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = (n ? n : 1) + n / 8;   // headroom: batches of a file differ a little in size
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct bml_ctx {
    bml_params p{};
    bml::LocParams lp{};
    hipStream_t stream = nullptr;
    size_t scan_lds = 0;
    uint32_t max_pairs_per_chunk = 0;
    // genome
    bool loaded = false;
    uint32_t n_buckets = 0;
    std::vector<uint32_t> h_bucket_len;
    DevBuf<uint8_t> genome, lut, pair_rc;
    DevBuf<uint64_t> bucket_start, cand_start;
    DevBuf<uint32_t> occ_a, samp_end;            // an occurrence is its offset in the bucket (4 bytes); per (candidate, sample): where its group ends
    DevBuf<uint32_t> bucket_len, sample_hash, seg_len, pair_window, out_votes;
    DevBuf<uint32_t> cand_count, heavy, n_heavy, heavy_votes, heavy_bitmaps;
    double occ_per_pair = 0;                     // k-mer occurrences per candidate the batches so far needed (sizes the next buffer)
    int n_cu = 256;
    uint32_t last_heavy = 0;
    DevBuf<uint16_t> sample_pos;
    DevBuf<int32_t> out_offset;
    DevBuf<bml::Chunk> chunks;
    // bml_sample_windows
    DevBuf<uint8_t> s_bases, s_quals, s_has;
    DevBuf<uint64_t> s_win_start;
    DevBuf<uint32_t> s_win_len, s_hash;
    DevBuf<uint16_t> s_pos, s_table;
    uint32_t s_table_len = 0;        // windows up to this length are tabulated
    // bml_sample_text_windows: two page-locked slots for the gathered windows, two for the results
    uint8_t *t_bases[2] = {nullptr, nullptr}, *t_quals[2] = {nullptr, nullptr}, *t_out[2] = {nullptr, nullptr};
    size_t t_cap[2] = {0, 0}, t_out_cap[2] = {0, 0};
    hipEvent_t t_done[2] = {nullptr, nullptr};
    DevBuf<unsigned long long> occ_count;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // scan | (gap) | light replay | heavy replay
    float ms[3] = {0, 0, 0};         // scan kernels | host time between and around the kernels | replay kernels
    uint64_t last_occ = 0;
};

// Sampler(p).sample_deterministically(n - 1) for every possible selection size n (utils.h:160-178), in
// fp64 on the host so that device rounding can never differ; tabulated for windows of up to max_len bases
static int sampler_table(bml_ctx *c, uint32_t max_len) {
    if (max_len <= c->s_table_len) return BML_OK;
    const uint32_t k = c->p.k, p = c->p.num_samples;
    HIP_TRY(hipStreamSynchronize(c->stream));                    // (a kernel in flight may still read the old table)
    const uint32_t max_nk = max_len >= k ? max_len - k + 1 : 0;
    std::vector<uint16_t> tab((size_t)(max_nk + 1) * p, 0);
    for (uint32_t n = 1; n <= max_nk; n++) {
        const uint32_t ub = n - 1;
        double delta = 0.0;
        if (p != 1) delta = (double)(ub + 1u) / (double)(p - 1u);
        for (uint32_t s = 0; s + 1 < p; s++) tab[(size_t)n * p + s] = (uint16_t)floor((double)s * delta);
        tab[(size_t)n * p + p - 1] = (uint16_t)ub;
    }
    HIP_TRY(c->s_table.need(tab.size()));
    HIP_TRY(hipMemcpy(c->s_table.p, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    c->s_table_len = max_len;
    return BML_OK;
}

extern "C" {

const char *bml_last_error(void) { return g_err; }

int bml_create(const bml_params *params, bml_ctx **out) {
    if (!params || !out) return fail(BML_ERR_ARG, "bml_create: null argument");
    *out = nullptr;
    const bml_params &p = *params;
    if (p.k == 0 || p.k > 16) return fail(BML_ERR_ARG, "k must be in 1..16 (got %u)", p.k);
    if (p.num_samples == 0 || p.num_samples > 64) return fail(BML_ERR_UNSUPPORTED, "num_samples must be in 1..64");
    if (p.max_bucket_bases == 0) return fail(BML_ERR_ARG, "max_bucket_bases must be > 0");
    if (p.max_bucket_bases >= (1u << 20)) return fail(BML_ERR_UNSUPPORTED, "buckets of 2^20 bases or more are not supported");   // 20-bit offsets in the light replay's sort keys
    const uint32_t max_words = (p.max_bucket_bases + 15u) / 16u;
    const uint32_t max_pairs = bml::kMaxTargets / p.num_samples;
    const size_t lds = bml::scan_lds_bytes(max_words);
    if (lds > 160 * 1024) return fail(BML_ERR_UNSUPPORTED, "buckets of %u bases do not fit the 160 KiB LDS", p.max_bucket_bases);
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (p.device < 0 || p.device >= n_dev) return fail(BML_ERR_HIP, "device %d not available (%d HIP devices)", p.device, n_dev);
    HIP_TRY(hipSetDevice(p.device));
    bml_ctx *c = new bml_ctx();
    c->p = p;
    c->lp.k = p.k;
    c->lp.p = p.num_samples;
    c->lp.allowed_mismatch = p.allowed_mismatch;
    c->lp.allowed_indel = p.allowed_indel;
    c->lp.max_words = max_words;
    c->lp.max_pairs = max_pairs;
    c->scan_lds = lds;
    c->max_pairs_per_chunk = max_pairs;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&c->ev[i]);
    if (e == hipSuccess && lds > 48 * 1024)
        e = bmhip::raise_dynamic_lds(reinterpret_cast<const void *>(bml::bml_scan_kernel), lds);
    uint8_t lut[256];
    build_dna4_lut(lut);
    if (e == hipSuccess) e = c->lut.need(256);
    if (e == hipSuccess) e = hipMemcpy(c->lut.p, lut, 256, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = c->occ_count.need(2);
    if (e == hipSuccess) e = c->n_heavy.need(3);
    if (e == hipSuccess) (void)hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, p.device);
    if (c->n_cu <= 0) c->n_cu = 256;
    if (e != hipSuccess) {
        bml_destroy(c);
        return fail(BML_ERR_HIP, "bml_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return BML_OK;
}

void bml_destroy(bml_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->genome.release(); c->lut.release(); c->pair_rc.release();
    c->bucket_start.release(); c->occ_a.release(); c->samp_end.release(); c->cand_start.release();
    c->cand_count.release(); c->heavy.release(); c->n_heavy.release(); c->heavy_votes.release();
    c->heavy_bitmaps.release();
    c->bucket_len.release(); c->sample_hash.release(); c->seg_len.release(); c->pair_window.release();
    c->out_votes.release(); c->sample_pos.release();
    c->out_offset.release(); c->chunks.release(); c->occ_count.release();
    c->s_bases.release(); c->s_quals.release(); c->s_has.release(); c->s_win_start.release(); c->s_win_len.release();
    c->s_hash.release(); c->s_pos.release(); c->s_table.release();
    for (int i = 0; i < 2; i++) {
        if (c->t_bases[i]) (void)hipHostFree(c->t_bases[i]);
        if (c->t_quals[i]) (void)hipHostFree(c->t_quals[i]);
        if (c->t_out[i]) (void)hipHostFree(c->t_out[i]);
        if (c->t_done[i]) (void)hipEventDestroy(c->t_done[i]);
    }
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bml_sample_windows(bml_ctx *c, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes, const uint64_t *win_start,
                       const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality, uint32_t *out_hash,
                       uint16_t *out_pos, uint8_t *out_has) {
    if (!c) return fail(BML_ERR_ARG, "bml_sample_windows: null context");
    if (n_windows == 0) return BML_OK;
    if (!win_start || !win_len || !out_hash || !out_pos || !out_has || (n_bytes && (!bases || !quals)))
        return fail(BML_ERR_ARG, "bml_sample_windows: null argument");
    if (n_windows >= (1u << 26)) return fail(BML_ERR_UNSUPPORTED, "too many windows in one call (%u)", n_windows);   // 64 threads each, < 2^32
    uint32_t max_len = 1;
    for (uint32_t w = 0; w < n_windows; w++) {
        if (win_start[w] > n_bytes || win_len[w] > n_bytes - win_start[w])
            return fail(BML_ERR_ARG, "window %u lies outside the read buffer", w);
        if (win_len[w] > 0xFFFFu) return fail(BML_ERR_UNSUPPORTED, "window %u is %u bases long (positions are 16-bit)", w, win_len[w]);
        max_len = std::max(max_len, win_len[w]);
    }
    const uint32_t k = c->p.k, p = c->p.num_samples;
    const size_t lds = bml::sample_lds_bytes(max_len, k);
    if (lds > 160 * 1024) return fail(BML_ERR_UNSUPPORTED, "windows of %u bases need %zu B of LDS", max_len, lds);
    HIP_TRY(hipSetDevice(c->p.device));
    if (lds > 48 * 1024)
        HIP_TRY(bmhip::raise_dynamic_lds(reinterpret_cast<const void *>(bml::bml_sample_kernel), lds));
    if (int rc = sampler_table(c, max_len)) return rc;
    HIP_TRY(c->s_bases.need((size_t)n_bytes));
    HIP_TRY(c->s_quals.need((size_t)n_bytes));
    HIP_TRY(c->s_win_start.need(n_windows));
    HIP_TRY(c->s_win_len.need(n_windows));
    HIP_TRY(c->s_hash.need((size_t)n_windows * p));
    HIP_TRY(c->s_pos.need((size_t)n_windows * p));
    HIP_TRY(c->s_has.need(n_windows));
    if (n_bytes) {
        HIP_TRY(hipMemcpyAsync(c->s_bases.p, bases, (size_t)n_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->s_quals.p, quals, (size_t)n_bytes, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->s_win_start.p, win_start, (size_t)n_windows * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->s_win_len.p, win_len, (size_t)n_windows * 4, hipMemcpyHostToDevice, c->stream));
    // the table is indexed by selection size for THIS table's p; sizes beyond the current windows are unused
    hipLaunchKernelGGL(bml::bml_sample_kernel, dim3(n_windows), dim3(64), lds, c->stream, k, p, min_base_quality, max_len,
                       c->s_bases.p, c->s_quals.p, c->s_win_start.p, c->s_win_len.p, c->lut.p, c->s_table.p, c->s_hash.p,
                       c->s_pos.p, c->s_has.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_hash, c->s_hash.p, (size_t)n_windows * p * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_pos, c->s_pos.p, (size_t)n_windows * p * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_has, c->s_has.p, (size_t)n_windows, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BML_OK;
}

// The same sampling for windows whose bases and qualities lie apart in one buffer (a FASTQ text): the library gathers a
// piece's windows back to back into page-locked buffers (a few threads) while the piece before is on the device.
int bml_sample_text_windows(bml_ctx *c, const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start, const uint64_t *qual_start,
                            const uint32_t *win_len, uint32_t n_windows, uint32_t min_base_quality, uint32_t *out_hash,
                            uint16_t *out_pos, uint8_t *out_has) {
    if (!c) return fail(BML_ERR_ARG, "bml_sample_text_windows: null context");
    if (n_windows == 0) return BML_OK;
    if (!seq_start || !qual_start || !win_len || !out_hash || !out_pos || !out_has || (n_bytes && !text))
        return fail(BML_ERR_ARG, "bml_sample_text_windows: null argument");
    uint32_t max_len = 1;
    for (uint32_t w = 0; w < n_windows; w++) {
        if (seq_start[w] > n_bytes || win_len[w] > n_bytes - seq_start[w] || qual_start[w] > n_bytes || win_len[w] > n_bytes - qual_start[w])
            return fail(BML_ERR_ARG, "window %u lies outside the text", w);
        if (win_len[w] > 0xFFFFu) return fail(BML_ERR_UNSUPPORTED, "window %u is %u bases long (positions are 16-bit)", w, win_len[w]);
        max_len = std::max(max_len, win_len[w]);
    }
    HIP_TRY(hipSetDevice(c->p.device));
    // pieces of about 32 MiB of bases; two page-locked slots in, two out
    const uint32_t piece = std::max<uint32_t>(1024u, (uint32_t)std::min<uint64_t>(1u << 20, ((uint64_t)32 << 20) / max_len));
    const uint32_t p = c->p.num_samples;
    const size_t res_bytes = ((size_t)piece * (p * 6u + 1u) + 7u) & ~(size_t)7;      // hash | pos | has, then the piece's window views:
    const size_t out_bytes = res_bytes + (size_t)piece * 12;                         // start u64[piece] | len u32[piece]
    for (int i = 0; i < 2; i++) {
        if (!c->t_done[i]) HIP_TRY(hipEventCreateWithFlags(&c->t_done[i], hipEventDisableTiming));
        if (c->t_cap[i] < (size_t)piece * max_len + 64 || c->t_out_cap[i] < out_bytes) {
            if (c->t_bases[i]) (void)hipHostFree(c->t_bases[i]);
            if (c->t_quals[i]) (void)hipHostFree(c->t_quals[i]);
            if (c->t_out[i]) (void)hipHostFree(c->t_out[i]);
            c->t_bases[i] = c->t_quals[i] = c->t_out[i] = nullptr;
            c->t_cap[i] = c->t_out_cap[i] = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->t_bases[i]), (size_t)piece * max_len + 64, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->t_quals[i]), (size_t)piece * max_len + 64, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->t_out[i]), out_bytes, hipHostMallocDefault));
            c->t_cap[i] = (size_t)piece * max_len + 64;
            c->t_out_cap[i] = out_bytes;
        }
    }
    // the device buffers at a piece's largest, once: the pieces follow each other on the stream, none may be re-allocated
    // under the one before
    if (int rc = sampler_table(c, max_len)) return rc;
    HIP_TRY(c->s_bases.need((size_t)piece * max_len + 64));
    HIP_TRY(c->s_quals.need((size_t)piece * max_len + 64));
    HIP_TRY(c->s_win_start.need(piece));
    HIP_TRY(c->s_win_len.need(piece));
    HIP_TRY(c->s_hash.need((size_t)piece * p));
    HIP_TRY(c->s_pos.need((size_t)piece * p));
    HIP_TRY(c->s_has.need(piece));
    const unsigned hw = std::max(1u, std::min(3u, std::thread::hardware_concurrency() / 2u));
    struct InFlight { uint32_t first = 0, n = 0; bool busy = false; } fl[2];
    auto finish = [&](int slot) -> int {                         // the piece's results, from the page-locked slot to the caller
        if (!fl[slot].busy) return BML_OK;
        HIP_TRY(hipEventSynchronize(c->t_done[slot]));
        const uint32_t first = fl[slot].first, n = fl[slot].n;
        const uint8_t *o = c->t_out[slot];
        memcpy(out_hash + (size_t)first * p, o, (size_t)n * p * 4);
        memcpy(out_pos + (size_t)first * p, o + (size_t)piece * p * 4, (size_t)n * p * 2);
        memcpy(out_has + first, o + (size_t)piece * p * 6, n);
        fl[slot].busy = false;
        return BML_OK;
    };
    uint32_t n_piece = 0;
    for (uint32_t first = 0; first < n_windows; first += piece, n_piece++) {
        const int slot = (int)(n_piece & 1);
        if (int rc = finish(slot)) return rc;                    // (also: the slot's upload has left its buffers)
        const uint32_t n = std::min(piece, n_windows - first);
        uint64_t *start = reinterpret_cast<uint64_t *>(c->t_out[slot] + res_bytes);
        uint32_t *lens = reinterpret_cast<uint32_t *>(c->t_out[slot] + res_bytes + (size_t)piece * 8);
        uint64_t at = 0;
        for (uint32_t w = 0; w < n; w++) {
            start[w] = at;
            lens[w] = win_len[first + w];
            at += lens[w];
        }
        auto gather = [&](uint32_t w0, uint32_t w1) {
            for (uint32_t w = w0; w < w1; w++) {
                memcpy(c->t_bases[slot] + start[w], text + seq_start[first + w], win_len[first + w]);
                memcpy(c->t_quals[slot] + start[w], text + qual_start[first + w], win_len[first + w]);
            }
        };
        const unsigned T = std::min(hw, std::max(1u, n / 2048u));
        if (T <= 1) {
            gather(0, n);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; t++) pool.emplace_back(gather, (uint32_t)((uint64_t)n * t / T), (uint32_t)((uint64_t)n * (t + 1) / T));
            gather(0, (uint32_t)((uint64_t)n / T));
            for (auto &t : pool) t.join();
        }
        // the rest is bml_sample_windows on the gathered piece, without a synchronisation
        const uint32_t k = c->p.k;
        const size_t lds = bml::sample_lds_bytes(max_len, k);
        if (lds > 160 * 1024) return fail(BML_ERR_UNSUPPORTED, "windows of %u bases need %zu B of LDS", max_len, lds);
        if (lds > 48 * 1024)
            HIP_TRY(bmhip::raise_dynamic_lds(reinterpret_cast<const void *>(bml::bml_sample_kernel), lds));
        if (at) {
            HIP_TRY(hipMemcpyAsync(c->s_bases.p, c->t_bases[slot], (size_t)at, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->s_quals.p, c->t_quals[slot], (size_t)at, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(c->s_win_start.p, start, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->s_win_len.p, lens, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(bml::bml_sample_kernel, dim3(n), dim3(64), lds, c->stream, k, p, min_base_quality, max_len, c->s_bases.p,
                           c->s_quals.p, c->s_win_start.p, c->s_win_len.p, c->lut.p, c->s_table.p, c->s_hash.p, c->s_pos.p, c->s_has.p);
        HIP_TRY(hipGetLastError());
        uint8_t *o = c->t_out[slot];
        HIP_TRY(hipMemcpyAsync(o, c->s_hash.p, (size_t)n * p * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(o + (size_t)piece * p * 4, c->s_pos.p, (size_t)n * p * 2, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(o + (size_t)piece * p * 6, c->s_has.p, n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->t_done[slot], c->stream));
        fl[slot].first = first;
        fl[slot].n = n;
        fl[slot].busy = true;
    }
    if (int rc = finish(0)) return rc;
    if (int rc = finish(1)) return rc;
    return BML_OK;
}

int bml_load_genome(bml_ctx *c, const uint8_t *bases, uint64_t n_bases, const uint64_t *bucket_start,
                    const uint32_t *bucket_len, uint32_t n_buckets) {
    if (!c || (n_bases && !bases) || (n_buckets && (!bucket_start || !bucket_len)))
        return fail(BML_ERR_ARG, "bml_load_genome: null argument");
    for (uint32_t b = 0; b < n_buckets; b++) {
        if (bucket_len[b] > c->p.max_bucket_bases)
            return fail(BML_ERR_ARG, "bucket %u has %u bases, more than max_bucket_bases = %u", b, bucket_len[b], c->p.max_bucket_bases);
        if (bucket_start[b] > n_bases || bucket_len[b] > n_bases - bucket_start[b])
            return fail(BML_ERR_ARG, "bucket %u lies outside the genome buffer", b);
    }
    HIP_TRY(hipSetDevice(c->p.device));
    // the scan kernel packs a bucket with aligned 16-byte loads, which read up to 15 bytes past its last base
    HIP_TRY(c->genome.need((size_t)n_bases + 64));
    HIP_TRY(hipMemset(c->genome.p + n_bases, 'A', 64));
    HIP_TRY(c->bucket_start.need(n_buckets));
    HIP_TRY(c->bucket_len.need(n_buckets));
    if (n_bases) HIP_TRY(bmhip::upload_pageable(c->genome.p, bases, (size_t)n_bases));
    if (n_buckets) {
        HIP_TRY(hipMemcpy(c->bucket_start.p, bucket_start, (size_t)n_buckets * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->bucket_len.p, bucket_len, (size_t)n_buckets * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    c->n_buckets = n_buckets;
    c->h_bucket_len.assign(bucket_len, bucket_len + n_buckets);
    c->loaded = true;
    return BML_OK;
}

// The same for a genome whose records are buffers of their own on the host (no flattened copy needed): the records go
// to the device back to back, in order; bucket_start counts in that concatenation.
int bml_load_genome_records(bml_ctx *c, const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records,
                            const uint64_t *bucket_start, const uint32_t *bucket_len, uint32_t n_buckets) {
    if (!c || (n_records && (!rec || !rec_len)) || (n_buckets && (!bucket_start || !bucket_len)))
        return fail(BML_ERR_ARG, "bml_load_genome_records: null argument");
    uint64_t n_bases = 0;
    for (uint32_t r = 0; r < n_records; r++) {
        if (rec_len[r] && !rec[r]) return fail(BML_ERR_ARG, "bml_load_genome_records: record %u is null", r);
        n_bases += rec_len[r];
    }
    for (uint32_t b = 0; b < n_buckets; b++) {
        if (bucket_len[b] > c->p.max_bucket_bases)
            return fail(BML_ERR_ARG, "bucket %u has %u bases, more than max_bucket_bases = %u", b, bucket_len[b], c->p.max_bucket_bases);
        if (bucket_start[b] > n_bases || bucket_len[b] > n_bases - bucket_start[b])
            return fail(BML_ERR_ARG, "bucket %u lies outside the genome buffer", b);
    }
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(c->genome.need((size_t)n_bases + 64));
    HIP_TRY(hipMemset(c->genome.p + n_bases, 'A', 64));
    HIP_TRY(c->bucket_start.need(n_buckets));
    HIP_TRY(c->bucket_len.need(n_buckets));
    HIP_TRY(bmhip::upload_pageable_records(c->genome.p, rec, rec_len, n_records));
    if (n_buckets) {
        HIP_TRY(hipMemcpy(c->bucket_start.p, bucket_start, (size_t)n_buckets * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->bucket_len.p, bucket_len, (size_t)n_buckets * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    c->n_buckets = n_buckets;
    c->h_bucket_len.assign(bucket_len, bucket_len + n_buckets);
    c->loaded = true;
    return BML_OK;
}

int bml_locate(bml_ctx *c, const uint32_t *sample_hash, const uint16_t *sample_pos, const uint32_t *seg_len,
               uint32_t n_windows, const uint32_t *pair_bucket, const uint32_t *pair_window, const uint8_t *pair_rc,
               uint32_t n_pairs, int32_t *out_offset, uint32_t *out_votes) {
    if (!c) return fail(BML_ERR_ARG, "bml_locate: null context");
    if (!c->loaded) return fail(BML_ERR_STATE, "no genome loaded");
    if (n_pairs == 0) return BML_OK;
    if (!sample_hash || !sample_pos || !seg_len || !pair_bucket || !pair_window || !pair_rc || !out_offset || !out_votes)
        return fail(BML_ERR_ARG, "bml_locate: null argument");
    const uint32_t p = c->p.num_samples;
    if ((uint64_t)n_pairs * p >= 0xFFFFFFFFull) return fail(BML_ERR_UNSUPPORTED, "too many candidates in one batch");
    // chunks: all candidates of a bucket are adjacent; split long runs so that a chunk's k-mers fit the LDS table
    std::vector<bml::Chunk> chunks;
    for (uint32_t i = 0; i < n_pairs;) {
        const uint32_t b = pair_bucket[i];
        if (b >= c->n_buckets) return fail(BML_ERR_ARG, "candidate %u names bucket %u, but only %u buckets are loaded", i, b, c->n_buckets);
        uint32_t j = i;
        while (j < n_pairs && pair_bucket[j] == b) j++;
        for (uint32_t s = i; s < j; s += c->max_pairs_per_chunk)
            chunks.push_back(bml::Chunk{b, s, (j - s < c->max_pairs_per_chunk) ? j - s : c->max_pairs_per_chunk});
        i = j;
    }
    if (chunks.size() >= (1u << 22)) return fail(BML_ERR_UNSUPPORTED, "too many candidate chunks in one call (%zu)", chunks.size());   // 1 024 threads each, < 2^32
    for (uint32_t i = 0; i < n_pairs; i++) {
        if (pair_window[i] >= n_windows) return fail(BML_ERR_ARG, "candidate %u names window %u of %u", i, pair_window[i], n_windows);
        // :242 `length - k - index` must not wrap for reverse-complement candidates
        if (seg_len[pair_window[i]] < c->p.k) return fail(BML_ERR_ARG, "candidate %u: window shorter than k", i);
    }
    HIP_TRY(hipSetDevice(c->p.device));
    const size_t n_s = (size_t)n_windows * p;
    HIP_TRY(c->sample_hash.need(n_s));
    HIP_TRY(c->sample_pos.need(n_s));
    HIP_TRY(c->seg_len.need(n_windows));
    HIP_TRY(c->pair_window.need(n_pairs));
    HIP_TRY(c->pair_rc.need(n_pairs));
    HIP_TRY(c->chunks.need(chunks.size()));
    HIP_TRY(c->out_offset.need(n_pairs));
    HIP_TRY(c->out_votes.need(n_pairs));
    HIP_TRY(hipMemcpyAsync(c->sample_hash.p, sample_hash, n_s * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sample_pos.p, sample_pos, n_s * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->seg_len.p, seg_len, (size_t)n_windows * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pair_window.p, pair_window, (size_t)n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pair_rc.p, pair_rc, (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->chunks.p, chunks.data(), chunks.size() * sizeof(bml::Chunk), hipMemcpyHostToDevice, c->stream));

    // Occurrence buffer.  A true match per sample is the common case, so the first scan runs with room for two per sample;
    // it counts every candidate's occurrences whether they fit or not.  If they did not fit, the candidates are cut into
    // GROUPS of whole chunks whose occurrences fit a budget (BML_MAX_OCC occurrences, default 2^31 = 8 GiB of offsets: reads
    // in repeats bring 1 000+ occurrences per candidate, 11 G for 10 M reads on the genome-like genome), and every group is
    // scanned again and replayed on its own.
    HIP_TRY(c->cand_count.need(n_pairs));
    HIP_TRY(c->samp_end.need((size_t)n_pairs * p));
    HIP_TRY(c->cand_start.need(n_pairs));
    HIP_TRY(c->heavy.need(n_pairs));
    unsigned long long budget = 1ull << 31;
    if (const char *env = getenv("BML_MAX_OCC")) budget = std::max<unsigned long long>(1024, strtoull(env, nullptr, 10));
    uint32_t max_seg = 0;
    for (uint32_t w = 0; w < n_windows; w++) max_seg = std::max(max_seg, seg_len[w]);
    c->ms[0] = c->ms[1] = c->ms[2] = 0.f;
    c->last_occ = 0;
    c->last_heavy = 0;

    const auto t_begin = std::chrono::steady_clock::now();
    HIP_TRY(hipMemsetAsync(c->occ_count.p, 0, 2 * sizeof(unsigned long long), c->stream));
    // placed: the candidates' segments are where the scan before this one put them (it found the buffer too small, or the
    // host placed a group's segments from its counts): the kernel recounts -- counting costs it one pass over the bucket,
    // not one step per occurrence -- checks its counts against the segments it was given, and writes
    auto scan = [&](size_t chunk_lo, size_t chunk_hi, unsigned long long cap, unsigned long long *n_occ, bool placed = false) -> int {
        HIP_TRY(c->occ_a.need((size_t)cap));
        if (!placed) HIP_TRY(hipMemsetAsync(c->occ_count.p, 0, sizeof(unsigned long long), c->stream));
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        hipLaunchKernelGGL(bml::bml_scan_kernel, dim3((unsigned)(chunk_hi - chunk_lo)), dim3(bml::kScanThreads), c->scan_lds, c->stream,
                           c->lp, c->genome.p, c->bucket_start.p, c->bucket_len.p, c->chunks.p + chunk_lo, c->sample_hash.p,
                           c->pair_window.p, c->pair_rc.p, c->occ_a.p, c->occ_count.p, cap, c->cand_start.p, c->cand_count.p, c->samp_end.p,
                           placed ? 1u : 0u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        unsigned long long state[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(state, c->occ_count.p, sizeof state, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        *n_occ = state[0];
        if (state[1]) return fail(BML_ERR_HIP, "a scan's recount disagrees with the segments it was given");
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        c->ms[0] += ms;
        return BML_OK;
    };
    // the vote of the candidates [pair_lo, pair_hi), whose occurrences (n_occ of them) are in occ_a
    auto replay = [&](uint32_t pair_lo, uint32_t pair_hi, unsigned long long n_occ) -> int {
        HIP_TRY(hipMemsetAsync(c->n_heavy.p, 0, 3 * sizeof(uint32_t), c->stream));   // count, largest, queue head
        HIP_TRY(hipEventRecord(c->ev[2], c->stream));
        const uint32_t count = pair_hi - pair_lo;
        auto light = [&](auto kernel, unsigned threads) {
            hipLaunchKernelGGL(kernel, dim3((count + threads - 1) / threads), dim3(threads), 0, c->stream, c->lp, c->occ_a.p,
                               c->cand_start.p, c->cand_count.p, c->samp_end.p, c->sample_pos.p, c->seg_len.p, c->pair_window.p, c->pair_rc.p,
                               pair_lo, pair_hi, c->out_offset.p, c->out_votes.p, c->heavy.p, c->n_heavy.p);
        };
        if (p <= 10) light(bml::bml_replay_light_kernel<16, 256>, 256);
        else if (p <= 24) light(bml::bml_replay_light_kernel<32, 256>, 256);
        else light(bml::bml_replay_light_kernel<64, 128>, 128);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[4], c->stream));
        // candidates with more occurrences than the light kernel takes (repeats): one workgroup each, dense bitmaps of the
        // start positions
        uint32_t heavy_info[2] = {0, 0};                      // their number, and the most occurrences one of them has
        HIP_TRY(hipMemcpyAsync(heavy_info, c->n_heavy.p, sizeof heavy_info, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const uint32_t n_heavy = heavy_info[0];
        if (getenv("BML_LOG"))
            fprintf(stderr, "[bml] candidates %u..%u, %llu occurrences; %u heavy candidates, the largest with %u occurrences\n", pair_lo,
                    pair_hi, (unsigned long long)n_occ, n_heavy, heavy_info[1]);
        c->last_heavy += n_heavy;
        if (n_heavy) {
            const uint32_t range = c->p.max_bucket_bases + max_seg, words = (range + 31u) / 32u;
            // The three bitmaps in LDS while two workgroups still fit a CU (3 x 8 KB at 65 536-base buckets: six workgroups a CU,
            // 20 ms per million reads in repeats against 52 ms with the bitmaps -- and their atomics -- in global memory); beyond
            // that, global scratch and eight workgroups a CU (3 x 34 KB at 262 144: 25 ms for configs[4]'s 20 000 reads against
            // 30 ms at one workgroup a CU).  BML_LDS_BITMAP_KB moves the limit (experiments).
            const size_t lds_limit = getenv("BML_LDS_BITMAP_KB") ? (size_t)atoi(getenv("BML_LDS_BITMAP_KB")) * 1024 : (size_t)64 * 1024;
            const bool in_lds = (size_t)3 * words * sizeof(uint32_t) <= lds_limit;
            const size_t lds = in_lds ? (size_t)3 * words * sizeof(uint32_t) : 16;
            if (lds > 48 * 1024)
                HIP_TRY(bmhip::raise_dynamic_lds(reinterpret_cast<const void *>(bml::bml_replay_heavy_kernel), lds));
            // as many workgroups as the CUs hold at once (LDS decides: 1 per CU at 262 144-base buckets, 6 at 65 536)
            const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (size_t)(156 * 1024) / (lds + 1024)));
            const unsigned grid = (unsigned)std::min<uint32_t>(n_heavy, per_cu * (uint32_t)c->n_cu);
            // a candidate makes at most one proposal per start position and at most one per occurrence
            const uint32_t vote_stride = std::min(range, heavy_info[1]);
            HIP_TRY(c->heavy_votes.need((size_t)grid * 2 * vote_stride));
            if (!in_lds) HIP_TRY(c->heavy_bitmaps.need((size_t)grid * 3 * words));
            const bml::HeavyScratch hs{c->heavy_votes.p, c->heavy_bitmaps.p, vote_stride};
            HIP_TRY(hipEventRecord(c->ev[5], c->stream));        // (the allocations above are not replay time)
            hipLaunchKernelGGL(bml::bml_replay_heavy_kernel, dim3(grid), dim3(bml::kThreads), lds, c->stream, c->lp, c->occ_a.p,
                               c->cand_start.p, c->cand_count.p, c->samp_end.p, c->sample_pos.p, c->seg_len.p, c->pair_window.p, c->pair_rc.p, c->heavy.p,
                               c->n_heavy.p, range, max_seg, in_lds ? 1u : 0u, hs, c->out_offset.p, c->out_votes.p);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[2], c->ev[4]));            // light replay ...
        c->ms[2] += ms;
        if (n_heavy) {                                                    // ... + heavy replay, without the host gap between them
            HIP_TRY(hipEventElapsedTime(&ms, c->ev[5], c->ev[3]));
            c->ms[2] += ms;
        }
        return BML_OK;
    };

    // The buffer's first size: two occurrences per sample asked for -- or what the context's previous batches needed per
    // candidate, a quarter more (a genome's repeats show in every batch; a buffer found too small costs a second scan).
    unsigned long long cap = std::max<unsigned long long>(1ull << 20, 2ull * n_pairs * p);
    cap = std::max<unsigned long long>(cap, (unsigned long long)(c->occ_per_pair * 1.25 * (double)n_pairs));
    if (const char *env = getenv("BML_FIRST_OCC")) cap = std::max<unsigned long long>(1, strtoull(env, nullptr, 10));   // (tests: force the second scan)
    cap = std::min(cap, budget);
    unsigned long long n_occ = 0;
    if (int rc = scan(0, chunks.size(), cap, &n_occ)) return rc;
    c->last_occ = n_occ;
    c->occ_per_pair = std::max(0.5 * c->occ_per_pair, (double)n_occ / (double)std::max<uint32_t>(n_pairs, 1u));
    if (n_occ > cap && n_occ <= budget) {                       // too small, but one buffer will do: grow it and scan again
        // (with the quarter of headroom the next batch's estimate will ask for: growing a 10 GB buffer a second time costs
        // hundreds of milliseconds of hipFree + hipMalloc)
        HIP_TRY(c->occ_a.need((size_t)std::min<unsigned long long>(budget, n_occ + n_occ / 4)));
        cap = n_occ;
        if (int rc = scan(0, chunks.size(), cap, &n_occ, true)) return rc;
    }
    if (n_occ <= cap) {
        if (int rc = replay(0, n_pairs, n_occ)) return rc;
    } else {
        // groups of whole chunks within the budget (a single chunk beyond it is a group of its own)
        std::vector<uint32_t> counts(n_pairs);
        HIP_TRY(hipMemcpy(counts.data(), c->cand_count.p, (size_t)n_pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
        size_t lo = 0;
        while (lo < chunks.size()) {
            unsigned long long sum = 0;
            size_t hi = lo;
            while (hi < chunks.size()) {
                unsigned long long cs = 0;
                for (uint32_t i = 0; i < chunks[hi].pair_count; i++) cs += counts[chunks[hi].pair_begin + i];
                if (hi > lo && sum + cs > budget) break;
                sum += cs;
                hi++;
            }
            // the first scan has counted every candidate's occurrences: the group's segments are placed from those counts
            // (one after the other, in candidate order) and its scan only writes
            const uint32_t pair_lo = chunks[lo].pair_begin, pair_hi = chunks[hi - 1].pair_begin + chunks[hi - 1].pair_count;
            std::vector<uint64_t> placed(pair_hi - pair_lo);
            unsigned long long at = 0;
            for (uint32_t i = pair_lo; i < pair_hi; i++) {
                placed[i - pair_lo] = at;
                at += counts[i];
            }
            if (at != sum) return fail(BML_ERR_HIP, "a group's occurrence count does not add up (%llu, %llu)", sum, at);
            HIP_TRY(hipMemcpyAsync(c->cand_start.p + pair_lo, placed.data(), placed.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));          // (`placed` is pageable memory and goes out of scope)
            unsigned long long got = 0;
            if (int rc = scan(lo, hi, std::max<unsigned long long>(sum, 1), &got, true)) return rc;
            if (int rc = replay(pair_lo, pair_hi, sum)) return rc;
            lo = hi;
        }
    }
    // what the call spent outside its kernels from the first scan on: syncs, the count downloads, buffer growth, group placement
    c->ms[1] = std::max(0.f, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count() - c->ms[0] - c->ms[2]);
    HIP_TRY(hipMemcpyAsync(out_offset, c->out_offset.p, (size_t)n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out_votes, c->out_votes.p, (size_t)n_pairs * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BML_OK;
}

int bml_last_stats(bml_ctx *c, float *ms_scan, float *ms_host, float *ms_replay, uint64_t *n_occurrences) {
    if (!c) return fail(BML_ERR_ARG, "bml_last_stats: null context");
    if (ms_scan) *ms_scan = c->ms[0];
    if (ms_host) *ms_host = c->ms[1];
    if (ms_replay) *ms_replay = c->ms[2];
    if (n_occurrences) *n_occurrences = c->last_occ;
    return BML_OK;
}

int bml_last_count_histogram(bml_ctx *c, uint32_t n_pairs, uint64_t *candidates, uint64_t *occurrences) {
    if (!c || !candidates || !occurrences) return fail(BML_ERR_ARG, "bml_last_count_histogram: null argument");
    for (int i = 0; i < 33; i++) candidates[i] = occurrences[i] = 0;
    if (n_pairs == 0 || n_pairs > c->cand_count.cap) return BML_OK;
    HIP_TRY(hipSetDevice(c->p.device));
    std::vector<uint32_t> counts(n_pairs);
    HIP_TRY(hipMemcpy(counts.data(), c->cand_count.p, (size_t)n_pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t v : counts) {
        const int b = v ? 32 - __builtin_clz(v) : 0;              // 0, then [2^(b-1), 2^b)
        candidates[b]++;
        occurrences[b] += v;
    }
    return BML_OK;
}

int bml_last_heavy_candidates(bml_ctx *c, uint32_t *n_heavy) {
    if (!c || !n_heavy) return fail(BML_ERR_ARG, "bml_last_heavy_candidates: null argument");
    *n_heavy = c->last_heavy;
    return BML_OK;
}

}  // extern "C"
