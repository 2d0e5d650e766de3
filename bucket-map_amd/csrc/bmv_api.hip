// bmv_api.hip -- C ABI (include/bmv.h) over the alignment-verification kernels in bmv_kernels.hip.h.
// Host side only: validation, choice of the kernel shape for the longest query of the batch, chunking so
// that the traceback bits of one chunk fit the scratch budget, offsets of the packed CIGAR output
// (bm_scan.hip.h's exclusive sum).
#include "bmv_kernels.hip.h"

namespace bmv {
// instantiated in bmv_variants.hip
extern template __global__ void bmv_align_kernel<4, 5, false>(Job);
extern template __global__ void bmv_align_kernel<4, 6, false>(Job);
extern template __global__ void bmv_align_kernel<4, 7, false>(Job);
extern template __global__ void bmv_align_kernel<4, 8, false>(Job);
extern template __global__ void bmv_align_kernel<8, 4, false>(Job);
extern template __global__ void bmv_align_kernel<8, 5, false>(Job);
extern template __global__ void bmv_align_kernel<8, 6, false>(Job);
extern template __global__ void bmv_align_kernel<8, 7, false>(Job);
extern template __global__ void bmv_align_kernel<8, 8, false>(Job);
extern template __global__ void bmv_align_kernel<4, 6, true>(Job);
extern template __global__ void bmv_align_kernel<4, 8, true>(Job);
extern template __global__ void bmv_align_lane_kernel<1>(Job);
extern template __global__ void bmv_align_lane_kernel<2>(Job);
extern template __global__ void bmv_align_lane_kernel<3>(Job);
extern template __global__ void bmv_align_lane_kernel<4>(Job);
extern template __global__ void bmv_align_lane_kernel<5>(Job);
extern template __global__ void bmv_align_lane_kernel<6>(Job);
extern template __global__ void bmv_align_lane_kernel<7>(Job);
extern template __global__ void bmv_align_lane_kernel<8>(Job);

// CIGAR entries of one chunk, reversed into reading order at their final offsets.
__global__ void bmv_gather_kernel(const uint32_t *__restrict__ ops_rev, uint32_t ops_stride,
                                  const uint32_t *__restrict__ nops, const uint32_t *__restrict__ offsets,
                                  uint32_t count, uint32_t *__restrict__ out) {
    const uint32_t slot = blockIdx.x * (blockDim.x / 8u) + threadIdx.x / 8u, t = threadIdx.x % 8u;
    if (slot >= count) return;
    const uint32_t c = nops[slot];
    const uint32_t *src = ops_rev + (size_t)slot * ops_stride;
    uint32_t *dst = out + offsets[slot];
    for (uint32_t x = t; x < c; x += 8u) dst[x] = src[c - 1u - x];
}

}  // namespace bmv
#include "bm_hip_util.h"

#include "../../include/bmv.h"

#include "bm_scan.hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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
            return fail(BMV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

void build_dna4_lut(uint8_t *lut) {
    memset(lut, 0, 256);
    const char *m[4] = {"AaRrWwMmDdHhVv", "CcYySsBb", "GgKk", "TtUu"};
    for (int r = 0; r < 4; r++)
        for (const char *c = m[r]; *c; c++) lut[(uint8_t)*c] = (uint8_t)r;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n) {
        if (n <= cap && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T));
        if (e == hipSuccess) cap = n ? n : 1;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

using align_fn = void (*)(bmv::Job);

struct Shape {
    uint32_t group;   // lanes per alignment
    int cw;           // 64-row words per lane
    align_fn fn;
    bool per_lane;    // bmv_align_lane_kernel: one alignment per lane (group 1), its own trace and text layout
};

// A group is as many lanes as the longest query has words, over CW words per lane; the kernel variant fixes CW and how many
// of the group's lanes hold the traceback's 16 pairs of trace words.  CW is chosen by what a wave then spends per
// alignment: the cost of a column step at that CW (the table below: what a step pays once -- the neighbour lane's delta,
// the text base, the loop -- is worth about two words' recurrences), times n + group - 1 steps (the skew), over the
// 64 / group alignments that share them.  A 5-kbp read (79 words) is 4 alignments of 16 lanes x 5 words per wave, not
// one of 40 x 2; a 10-kbp read (157 words) is 2 of 32 x 5, not one of 40 x 4; an 8-kbp read (125 words) 4 of 16 x 8.
constexpr int kMaxCw = 8;
// what one column step of a wave costs at CW words per lane, measured (ns of the whole card per wave and step,
// tools/bench_verify_cw.sh, profiles/r02/verify_shapes.txt; only the ratios matter): about 1.9 + CW words' recurrences --
// what a step pays once (the neighbour lane's delta, the text base, the loop) is worth two words
constexpr double kStepCost[kMaxCw + 1] = {0, 0.206, 0.290, 0.356, 0.440, 0.545, 0.645, 0.700, 0.810};
constexpr uint32_t kStripsBeyond = 64u * kMaxCw;                 // words: longer queries go through in strips

// Queries of up to this many words go one per lane (bmv_align_lane_kernel): 64 alignments share every instruction of a
// column step.  BMV_LANE_MAX=0 switches it off (experiments; tests run both ways).
constexpr uint32_t kLaneWords = 8;
constexpr uint32_t kLaneMaxText = 2048;

Shape pick_shape(uint32_t words, uint32_t max_n) {
    // (64 text windows in LDS, 2 bits a base: 33 KB at 2 048 bases -- longer windows would leave a CU fewer than four waves)
    if (max_n <= kLaneMaxText) {
        static const align_fn per_lane[kLaneWords + 1] = {nullptr,
                                                          bmv::bmv_align_lane_kernel<1>, bmv::bmv_align_lane_kernel<2>,
                                                          bmv::bmv_align_lane_kernel<3>, bmv::bmv_align_lane_kernel<4>,
                                                          bmv::bmv_align_lane_kernel<5>, bmv::bmv_align_lane_kernel<6>,
                                                          bmv::bmv_align_lane_kernel<7>, bmv::bmv_align_lane_kernel<8>};
        const char *env = getenv("BMV_LANE_MAX");
        const uint32_t lane_max = env ? std::min<uint32_t>((uint32_t)atoi(env), kLaneWords) : kLaneWords;
        if (words <= lane_max) return Shape{1u, (int)words, per_lane[words], true};
    }
    // strips of 64 * CW words, one after the other (max_query_len = 65 536 bases: two of them)
    if (words > kStripsBeyond)
        return words <= 2u * 64u * 6u ? Shape{64u, 6, bmv::bmv_align_kernel<4, 6, true>, false} : Shape{64u, 8, bmv::bmv_align_kernel<4, 8, true>, false};
    // A group's lanes share the traceback's trace words, 16 / SLOTS lanes to a (word, block) cell and SLOTS columns each, and
    // a round of the traceback recomputes as many cells as the group has lanes for: groups of 8 lanes and more use the
    // SLOTS = 4 kernels (GROUP / 4 cells), groups of 2..7 lanes the SLOTS = 8 ones (GROUP / 2 cells: 1 kbp as 4 lanes x 4
    // words 12.7 -> 11.8 ms, 600 bases as 2 x 5 instead of 4 x 3 8.6 -> 6.7 ms), a lone lane SLOTS = 16 (CW = 1; only with
    // BMV_LANE_MAX=0 or a text window too long for the lane kernel).
    // (experiment knob; never below 4: a cell of the SLOTS = 4 kernels needs four lanes)
    const uint32_t kEightBelow = getenv("BMV_EIGHT_BELOW") ? std::max(4u, (uint32_t)atoi(getenv("BMV_EIGHT_BELOW"))) : 8u;
    static const align_fn four_cols[kMaxCw + 1] = {nullptr,
                                                   bmv::bmv_align_kernel<4, 1, false>, bmv::bmv_align_kernel<4, 2, false>,
                                                   bmv::bmv_align_kernel<4, 3, false>, bmv::bmv_align_kernel<4, 4, false>,
                                                   bmv::bmv_align_kernel<4, 5, false>, bmv::bmv_align_kernel<4, 6, false>,
                                                   bmv::bmv_align_kernel<4, 7, false>, bmv::bmv_align_kernel<4, 8, false>};
    static const align_fn eight_cols[kMaxCw + 1] = {nullptr,
                                                    bmv::bmv_align_kernel<8, 1, false>, bmv::bmv_align_kernel<8, 2, false>,
                                                    bmv::bmv_align_kernel<8, 3, false>, bmv::bmv_align_kernel<8, 4, false>,
                                                    bmv::bmv_align_kernel<8, 5, false>, bmv::bmv_align_kernel<8, 6, false>,
                                                    bmv::bmv_align_kernel<8, 7, false>, bmv::bmv_align_kernel<8, 8, false>};
    int cw = 0;
    double best = 0;
    const char *env = getenv("BMV_CW");                          // experiment / test knob: force CW where it is possible
    const int forced = env ? atoi(env) : 0;
    for (int c = 1; c <= kMaxCw; c++) {
        const uint32_t g = (words + (uint32_t)c - 1u) / (uint32_t)c;
        const int max_cw = g >= 2u ? kMaxCw : 1;
        if (g > 64u || c > max_cw) continue;
        if (forced == c) {
            cw = c;
            break;
        }
        const double cost = kStepCost[c] * (double)(max_n + g - 1u) / (double)(64u / g);
        if (cw == 0 || cost < best * 0.98) {
            best = cost;
            cw = c;
        }
    }
    const uint32_t g = std::max(1u, (words + (uint32_t)cw - 1u) / (uint32_t)cw);
    if (g >= kEightBelow) return {g, cw, four_cols[cw], false};
    if (g >= 2) return {g, cw, eight_cols[cw], false};
    return {g, 1, bmv::bmv_align_kernel<16, 1, false>, false};
}

}  // namespace

constexpr uint32_t kSideStreams = 4;

struct bmv_ctx {
    bmv_params p{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t side[kSideStreams] = {};       // length classes of one batch run side by side on these
    hipEvent_t side_done[kSideStreams] = {};
    bool loaded = false;
    uint64_t n_genome = 0;
    size_t scratch_bytes = 0;
    DevBuf<uint8_t> genome, lut, reads, text_rc, scan_tmp;
    DevBuf<uint64_t> text_start, query_start, trace;
    DevBuf<uint32_t> text_len, query_len, ops_rev, nops, offsets, packed, out_begin, order;
    DevBuf<int32_t> out_score;
    // results of the last bmv_align, host side
    uint32_t n_last = 0;
    std::vector<int32_t> h_score;
    std::vector<uint32_t> h_begin, h_cigar;
    std::vector<uint64_t> h_offset;
    float ms_kernels = 0.f;
    uint64_t n_cells = 0;
};

extern "C" {

const char *bmv_last_error(void) { return g_err; }

int bmv_create(const bmv_params *params, bmv_ctx **out) {
    if (!params || !out) return fail(BMV_ERR_ARG, "bmv_create: null argument");
    *out = nullptr;
    if (params->max_query_len == 0 || params->max_query_len > 65536)
        return fail(BMV_ERR_UNSUPPORTED, "max_query_len must be in 1..65536 (got %u)", params->max_query_len);
    if (params->max_text_len == 0 || params->max_text_len > 81920)
        return fail(BMV_ERR_UNSUPPORTED, "max_text_len must be in 1..81920 (got %u)", params->max_text_len);
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (params->device < 0 || params->device >= n_dev)
        return fail(BMV_ERR_HIP, "device %d not available (%d HIP devices)", params->device, n_dev);
    HIP_TRY(hipSetDevice(params->device));
    bmv_ctx *c = new bmv_ctx();
    c->p = *params;
    // checkpoints of the alignments in flight: a third of the free HBM, 1..96 GiB (a 30-kbp alignment keeps 19 MB and
    // is one wave: long reads want thousands in flight, and a mixed batch wants several length classes in flight
    // together -- 40 000 alignments of 1 .. 30 kbp: 26.8 / 31.2 / 32.8 / 35.0 T cell updates/s with 48 / 72 / 96 / 140 GB);
    // BMV_SCRATCH_MB overrides (tests use it to force chunking)
    size_t free_b = 0, total_b = 0;
    c->scratch_bytes = (size_t)8 << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        c->scratch_bytes = std::min<size_t>(std::max<size_t>(free_b / 3, (size_t)1 << 30), (size_t)96 << 30);
    if (const char *env = getenv("BMV_SCRATCH_MB")) {
        const long v = strtol(env, nullptr, 10);
        if (v > 0) c->scratch_bytes = (size_t)v << 20;
    }
    uint8_t lut[256];
    build_dna4_lut(lut);
    bool side_ok = true;
    for (uint32_t k = 0; k < kSideStreams; k++)
        side_ok = side_ok && hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&c->side_done[k], hipEventDisableTiming) == hipSuccess;
    if (!side_ok || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        c->lut.need(256) != hipSuccess || hipMemcpy(c->lut.p, lut, 256, hipMemcpyHostToDevice) != hipSuccess) {
        bmv_destroy(c);
        return fail(BMV_ERR_HIP, "bmv_create: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = c;
    return BMV_OK;
}

void bmv_destroy(bmv_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (uint32_t k = 0; k < kSideStreams; k++) {
        if (c->side[k]) {
            (void)hipStreamSynchronize(c->side[k]);
            (void)hipStreamDestroy(c->side[k]);
        }
        if (c->side_done[k]) (void)hipEventDestroy(c->side_done[k]);
    }
    c->genome.release(); c->lut.release(); c->reads.release(); c->text_rc.release(); c->scan_tmp.release();
    c->text_start.release(); c->query_start.release(); c->trace.release();
    c->text_len.release(); c->query_len.release(); c->ops_rev.release(); c->nops.release(); c->offsets.release();
    c->packed.release(); c->out_begin.release(); c->out_score.release(); c->order.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bmv_load_genome(bmv_ctx *c, const uint8_t *bases, uint64_t n_bases) {
    if (!c || (n_bases && !bases)) return fail(BMV_ERR_ARG, "bmv_load_genome: null argument");
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(c->genome.need((size_t)n_bases + 64u));             // (slack: an empty text window at the very end is still fetched from)
    if (n_bases) HIP_TRY(bmhip::upload_pageable(c->genome.p, bases, (size_t)n_bases));
    c->n_genome = n_bases;
    c->loaded = true;
    return BMV_OK;
}

int bmv_load_genome_records(bmv_ctx *c, const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records) {
    if (!c || (n_records && (!rec || !rec_len))) return fail(BMV_ERR_ARG, "bmv_load_genome_records: null argument");
    uint64_t n_bases = 0;
    for (uint32_t r = 0; r < n_records; r++) {
        if (rec_len[r] && !rec[r]) return fail(BMV_ERR_ARG, "bmv_load_genome_records: record %u is null", r);
        n_bases += rec_len[r];
    }
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(c->genome.need((size_t)n_bases + 64u));
    HIP_TRY(bmhip::upload_pageable_records(c->genome.p, rec, rec_len, n_records));
    c->n_genome = n_bases;
    c->loaded = true;
    return BMV_OK;
}

int bmv_align(bmv_ctx *c, const uint8_t *reads, uint64_t n_read_bytes, const uint64_t *text_start,
              const uint32_t *text_len, const uint8_t *text_rc, const uint64_t *query_start, const uint32_t *query_len,
              uint32_t n, uint64_t *total_cigar) {
    if (!c || !total_cigar) return fail(BMV_ERR_ARG, "bmv_align: null argument");
    if (!c->loaded) return fail(BMV_ERR_STATE, "bmv_align before bmv_load_genome");
    if (n && (!text_start || !text_len || !text_rc || !query_start || !query_len || (n_read_bytes && !reads)))
        return fail(BMV_ERR_ARG, "bmv_align: null argument");
    uint64_t cells = 0;
    for (uint32_t a = 0; a < n; a++) {
        if (query_len[a] > c->p.max_query_len)
            return fail(BMV_ERR_ARG, "alignment %u: query of %u bases, max_query_len is %u", a, query_len[a], c->p.max_query_len);
        if (text_len[a] > c->p.max_text_len)
            return fail(BMV_ERR_ARG, "alignment %u: text of %u bases, max_text_len is %u", a, text_len[a], c->p.max_text_len);
        if (query_start[a] > n_read_bytes || query_len[a] > n_read_bytes - query_start[a])
            return fail(BMV_ERR_ARG, "alignment %u: query lies outside the read buffer", a);
        if (text_start[a] > c->n_genome || text_len[a] > c->n_genome - text_start[a])
            return fail(BMV_ERR_ARG, "alignment %u: text lies outside the genome", a);
        cells += (uint64_t)query_len[a] * text_len[a];
    }
    c->n_last = n;
    c->n_cells = cells;
    c->ms_kernels = 0.f;
    c->h_score.assign(n, 0);
    c->h_begin.assign(n, 0);
    c->h_offset.assign((size_t)n + 1, 0);
    c->h_cigar.clear();
    *total_cigar = 0;
    if (n == 0) return BMV_OK;

    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(c->reads.need((size_t)n_read_bytes + 64u));         // (slack: so is an empty query)
    HIP_TRY(c->text_start.need(n));
    HIP_TRY(c->text_len.need(n));
    HIP_TRY(c->text_rc.need(n));
    HIP_TRY(c->query_start.need(n));
    HIP_TRY(c->query_len.need(n));
    HIP_TRY(c->out_score.need(n));
    HIP_TRY(c->out_begin.need(n));
    if (n_read_bytes) HIP_TRY(hipMemcpyAsync(c->reads.p, reads, (size_t)n_read_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->text_start.p, text_start, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->text_len.p, text_len, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->text_rc.p, text_rc, (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->query_start.p, query_start, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->query_len.p, query_len, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));

    // Length classes.  The kernel's shape -- lanes per alignment, words per lane -- is fixed per launch by the longest query
    // it holds, and a 5-kbp read run in the shape of a 30-kbp one costs six times what it should: the batch is cut into
    // classes of similar query length (in 64-row words; everything up to 8 words -- 512 bases, one alignment per lane -- is
    // one class, so a short-read batch stays one launch; then up to 16 words), each class is a launch series of its own over an index list, and the
    // results find their way back through that list.
    static const uint32_t kClassUpTo[] = {kLaneWords, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1024};
    constexpr uint32_t kClasses = sizeof kClassUpTo / sizeof kClassUpTo[0];
    const bool classes_off = getenv("BMV_ONE_CLASS") != nullptr;    // experiment knob: the whole batch in the longest query's shape
    auto class_of = [&](uint32_t m) {
        const uint32_t words = classes_off ? 0u : (m + 63u) / 64u;
        uint32_t k = 0;
        while (k + 1u < kClasses && words > kClassUpTo[k]) k++;
        return k;
    };
    uint32_t class_lo[kClasses + 1] = {0};
    for (uint32_t a = 0; a < n; a++) class_lo[class_of(query_len[a]) + 1u]++;
    uint32_t n_classes = 0;
    for (uint32_t k = 0; k < kClasses; k++) {
        n_classes += class_lo[k + 1] ? 1u : 0u;
        class_lo[k + 1] += class_lo[k];
    }
    std::vector<uint32_t> order(n);
    {
        uint32_t at[kClasses];
        for (uint32_t k = 0; k < kClasses; k++) at[k] = class_lo[k];
        for (uint32_t a = 0; a < n; a++) order[at[class_of(query_len[a])]++] = a;      // stable: file order within a class
    }
    const bool one_class = n_classes == 1;                      // then order is the identity and the CIGARs arrive in place
    HIP_TRY(c->order.need(n));
    HIP_TRY(hipMemcpyAsync(c->order.p, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    std::vector<uint32_t> h_nops, h_packed, cigar_len, stash;
    std::vector<uint64_t> stash_at;
    if (!one_class) {
        cigar_len.assign(n, 0);
        stash_at.assign(n, 0);
    }
    // Grow these with headroom: the longest read differs a little from call to call, and a reallocation is a free, an
    // allocation and a synchronisation of the device each time (the calls themselves are cheap: 0.2-0.5 ms for 45 GiB,
    // tools/malloc_probe.py).
    auto with_headroom = [](size_t need, size_t have) { return need <= have ? have : need + need / 4; };

    // what each class needs, then one allocation for the largest of them
    struct Plan {
        uint32_t lo, members, max_m, trace_words, gpw, n_blocks, ops_stride, lds_stride;
        Shape sh;
        uint64_t trace_stride, chunk;
        size_t lds;
    };
    std::vector<Plan> plans;
    size_t need_trace = 0, need_ops = 0, need_slots = 0;
    // A batch of one length class (or of a few neighbouring ones: 10-kbp reads with indels straddle a class boundary) gains
    // nothing from more alignments in flight than fill the card, and fresh device memory is not free (the first call that
    // grew the scratch to 90 GB spent 2 s in hipMalloc): only a batch that really mixes lengths -- three classes or more,
    // the longest query at least twice the shortest class's -- may use more than 48 GiB.
    size_t budget = std::min<size_t>(c->scratch_bytes, (size_t)48 << 30);
    {
        uint32_t lo_m = 0xFFFFFFFFu, hi_m = 0;
        for (uint32_t a = 0; a < n; a++) {
            lo_m = std::min(lo_m, std::max(query_len[a], 1024u));
            hi_m = std::max(hi_m, query_len[a]);
        }
        if (n_classes >= 3 && hi_m >= 2u * lo_m) budget = c->scratch_bytes;
    }
    for (uint32_t k = 0; k < kClasses; k++) {
        Plan pl{};
        pl.lo = class_lo[k];
        pl.members = class_lo[k + 1] - pl.lo;
        if (pl.members == 0) continue;
        uint32_t max_m = 0, max_n = 0;
        for (uint32_t s0 = pl.lo; s0 < pl.lo + pl.members; s0++) {
            max_m = std::max(max_m, query_len[order[s0]]);
            max_n = std::max(max_n, text_len[order[s0]]);
        }
        pl.max_m = max_m;
        // kernel shape for the longest query of the class; scratch per alignment slot
        const uint32_t words = (max_m + 63u) / 64u;
        pl.sh = pick_shape(words ? words : 1u, max_n);
        pl.trace_words = std::max(std::max(words, 1u), pl.sh.group * (uint32_t)pl.sh.cw);   // (more than the words only for strips)
        pl.gpw = 64u / pl.sh.group;
        // checkpoints every 16 columns: (Pv, Mv) per word, plus one 32-bit word of horizontal deltas per word and block
        pl.n_blocks = (max_n + pl.sh.group + 15u) / 16u + 1u;     // blocks of 16 STEPS: the group's last lane is group - 1 steps behind
        const uint64_t n_entries = (uint64_t)pl.n_blocks * pl.gpw * pl.trace_words;
        pl.trace_stride = n_entries * 2u + (n_entries + 1u) / 2u;               // 64-bit words per wave
        pl.trace_stride += 2u * pl.trace_words * pl.gpw;                        // ... and the query's bit planes, for the traceback
        pl.ops_stride = max_m + max_n + 1u;
        pl.lds_stride = (max_n + 15u) / 16u * 4u + 4u;                          // the text as a 2-bit stream
        if (pl.sh.per_lane) pl.lds_stride = (max_n + 63u) / 64u * 16u + 8u;     // ... as two bit planes, 64 columns at a time
        pl.lds = 256 + (size_t)pl.gpw * pl.lds_stride;
        if (pl.lds > 160 * 1024) return fail(BMV_ERR_UNSUPPORTED, "text windows of %u bases need %zu B of LDS", max_n, pl.lds);
        if (pl.lds > 48 * 1024) HIP_TRY(bmhip::raise_dynamic_lds(reinterpret_cast<const void *>(pl.sh.fn), pl.lds));
        const uint64_t per_slot = pl.trace_stride / pl.gpw * 8u + (uint64_t)pl.ops_stride * 4u;
        uint64_t chunk = std::max<uint64_t>(pl.gpw, budget / per_slot);
        chunk = std::min<uint64_t>(chunk, (uint64_t)pl.gpw << 25);   // one wave per gpw alignments: waves * 64 threads < 2^32
        chunk = std::min<uint64_t>(chunk / pl.gpw * pl.gpw, (uint64_t)pl.members);
        if (chunk == 0) chunk = pl.members;
        // pieces of equal size: 40 000 alignments under a limit of 31 000 are 2 x 20 000, not 31 000 + 9 000 (a piece's
        // last waves run on a card that is emptying, whatever its size)
        const uint64_t n_pieces = (pl.members + chunk - 1u) / chunk;
        const uint64_t even = ((pl.members + n_pieces - 1u) / n_pieces + pl.gpw - 1u) / pl.gpw * pl.gpw;
        pl.chunk = std::min(chunk, even);
        need_trace = std::max(need_trace, (size_t)((pl.chunk + pl.gpw - 1u) / pl.gpw * pl.trace_stride));
        need_ops = std::max(need_ops, (size_t)(pl.chunk * pl.ops_stride));
        need_slots = std::max(need_slots, (size_t)pl.chunk);
        plans.push_back(pl);
    }
    const uint32_t stop_after = getenv("BMV_STOP_AFTER") ? (uint32_t)atoi(getenv("BMV_STOP_AFTER")) : 0u;   // phase timing
    auto launch = [&](const Plan &pl, uint64_t first, uint32_t count, hipStream_t stream, uint64_t *trace, uint32_t *ops_rev,
                      uint32_t *nops) {
        bmv::Job j{};
        j.genome = c->genome.p;
        j.reads = c->reads.p;
        j.lut = c->lut.p;
        j.text_start = c->text_start.p;
        j.text_len = c->text_len.p;
        j.text_rc = c->text_rc.p;
        j.query_start = c->query_start.p;
        j.query_len = c->query_len.p;
        j.order = c->order.p + pl.lo + first;
        j.count = count;
        j.trace = trace;
        j.trace_stride = pl.trace_stride;
        j.trace_words = pl.trace_words;
        j.trace_blocks = pl.n_blocks;
        j.group = pl.sh.group;
        j.ops_rev = ops_rev;
        j.ops_stride = pl.ops_stride;
        j.text_lds_stride = pl.lds_stride;
        j.out_score = c->out_score.p;
        j.out_begin = c->out_begin.p;
        j.out_nops = nops;
        j.stop_after = stop_after;
        hipLaunchKernelGGL(pl.sh.fn, dim3((count + pl.gpw - 1u) / pl.gpw), dim3(bmv::kWave), pl.lds, stream, j);
        return hipGetLastError();
    };
    // CIGARs of `count` slots (launch order) -> the host, through one exclusive sum and one gather per ops_stride
    struct Piece {
        const Plan *pl;
        uint64_t first;         // within the class
        uint32_t count, slot0;  // slots [slot0, slot0 + count) of nops / offsets
        uint32_t *ops_rev;
    };
    auto collect = [&](const std::vector<Piece> &pieces, uint32_t slots) -> int {
        HIP_TRY(c->scan_tmp.need(bmscan::tmp_elems(slots) * sizeof(uint32_t)));
        HIP_TRY(bmscan::exclusive_sum<uint32_t>(c->nops.p, c->offsets.p, slots, reinterpret_cast<uint32_t *>(c->scan_tmp.p), c->stream));
        uint32_t total = 0;                                     // the scan writes slots + 1 values: the last is the total
        HIP_TRY(hipMemcpyAsync(&total, c->offsets.p + slots, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(c->packed.need(with_headroom(total, c->packed.cap)));
        for (const Piece &pc : pieces) {
            hipLaunchKernelGGL(bmv::bmv_gather_kernel, dim3((pc.count + 31u) / 32u), dim3(256), 0, c->stream, pc.ops_rev,
                               pc.pl->ops_stride, c->nops.p + pc.slot0, c->offsets.p + pc.slot0, pc.count, c->packed.p);
            HIP_TRY(hipGetLastError());
        }
        h_nops.resize(slots);
        h_packed.resize(total);
        HIP_TRY(hipMemcpyAsync(h_nops.data(), c->nops.p, (size_t)slots * 4, hipMemcpyDeviceToHost, c->stream));
        if (total) HIP_TRY(hipMemcpyAsync(h_packed.data(), c->packed.p, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<uint32_t> &dst = one_class ? c->h_cigar : stash;
        uint64_t at = dst.size();
        for (const Piece &pc : pieces) {                        // in slot order, and the slots are theirs back to back
            for (uint32_t s0 = 0; s0 < pc.count; s0++) {
                const uint32_t len = h_nops[pc.slot0 + s0];
                if (one_class) {                                // order is the identity: the CIGARs arrive in place
                    c->h_offset[pc.first + s0] = at;
                } else {                                        // kept aside in launch order until every length is known
                    const uint32_t a = order[pc.pl->lo + pc.first + s0];
                    cigar_len[a] = len;
                    stash_at[a] = at;
                }
                at += len;
            }
        }
        dst.insert(dst.end(), h_packed.begin(), h_packed.end());
        return BMV_OK;
    };

    // Rounds.  A launch's scratch (checkpoints + reversed CIGAR entries) is what bounds how many alignments are in flight:
    // 19 MB per 30-kbp alignment.  A class of long reads is few waves that each run for tens of milliseconds, so one class
    // at a time leaves most of the card idle (measured on 40 000 alignments of 1 .. 30 kbp: 19.8 T cell updates/s against
    // 37-41 T for uniform batches).  Every class is therefore cut into pieces of at most a THIRD of the scratch budget (when
    // there is more than one class), the pieces -- longest reads first -- are packed into rounds that fit the budget
    // together, and the pieces of a round run side by side on a few streams; their CIGARs are collected once per round.
    // BMV_SERIAL_CLASSES=1: one piece per round (the old behaviour), for comparison.
    struct Todo {
        const Plan *pl;
        uint64_t first;
        uint32_t count;
        size_t trace_words, ops_words;
    };
    std::vector<Todo> todo;
    const bool serial = getenv("BMV_SERIAL_CLASSES") != nullptr;
    for (size_t i = plans.size(); i-- > 0;) {                   // the longest first: they take the longest
        const Plan &pl = plans[i];
        const uint64_t per_slot = pl.trace_stride / pl.gpw * 8u + (uint64_t)pl.ops_stride * 4u;
        uint64_t chunk = pl.chunk;
        if (plans.size() > 1 && !serial) {
            const uint64_t third = std::max<uint64_t>(pl.gpw, budget / 3u / per_slot) / pl.gpw * pl.gpw;
            chunk = std::min<uint64_t>(chunk, std::max<uint64_t>(third, pl.gpw));
            const uint64_t n_pieces = (pl.members + chunk - 1u) / chunk;
            chunk = std::min<uint64_t>(chunk, ((pl.members + n_pieces - 1u) / n_pieces + pl.gpw - 1u) / pl.gpw * pl.gpw);
        }
        for (uint64_t first = 0; first < pl.members; first += chunk) {
            const uint32_t count = (uint32_t)std::min<uint64_t>(chunk, pl.members - first);
            todo.push_back({&pl, first, count, (size_t)((count + pl.gpw - 1u) / pl.gpw * pl.trace_stride), (size_t)count * pl.ops_stride});
        }
    }
    for (size_t at = 0; at < todo.size();) {
        // the pieces of this round: as many as fit the budget (at least one)
        size_t end = at, sum_trace = 0, sum_ops = 0, slots = 0;
        while (end < todo.size() && (end == at || (!serial && (sum_trace + todo[end].trace_words) * 8u + (sum_ops + todo[end].ops_words) * 4u <=
                                                                   budget))) {
            sum_trace += todo[end].trace_words;
            sum_ops += todo[end].ops_words;
            slots += todo[end].count;
            end++;
        }
        HIP_TRY(c->trace.need(with_headroom(sum_trace, c->trace.cap)));
        HIP_TRY(c->ops_rev.need(with_headroom(sum_ops, c->ops_rev.cap)));
        HIP_TRY(c->nops.need(with_headroom(slots, c->nops.cap)));
        HIP_TRY(c->offsets.need(with_headroom(slots + 1, c->offsets.cap)));
        HIP_TRY(hipEventRecord(c->ev0, c->stream));             // the uploads above / the round before
        std::vector<Piece> pieces;
        size_t trace_at = 0, ops_at = 0;
        uint32_t slot_at = 0;
        const bool alone = end - at == 1;
        for (size_t i = at; i < end; i++) {
            const Todo &t = todo[i];
            hipStream_t st = alone ? c->stream : c->side[(i - at) % kSideStreams];
            if (!alone) HIP_TRY(hipStreamWaitEvent(st, c->ev0, 0));
            HIP_TRY(launch(*t.pl, t.first, t.count, st, c->trace.p + trace_at, c->ops_rev.p + ops_at, c->nops.p + slot_at));
            pieces.push_back({t.pl, t.first, t.count, slot_at, c->ops_rev.p + ops_at});
            trace_at += t.trace_words;
            ops_at += t.ops_words;
            slot_at += t.count;
        }
        if (!alone)
            for (uint32_t k = 0; k < kSideStreams; k++) {       // the main stream goes on when all of them are done
                HIP_TRY(hipEventRecord(c->side_done[k], c->side[k]));
                HIP_TRY(hipStreamWaitEvent(c->stream, c->side_done[k], 0));
            }
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        if (int rc = collect(pieces, slot_at)) return rc;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->ms_kernels += ms;
        if (getenv("BMV_LOG_CLASSES")) {
            fprintf(stderr, "[bmv] round of %zu piece(s), %.2f ms:", end - at, ms);
            for (size_t i = at; i < end; i++)
                fprintf(stderr, " [up to %u bases: %u of %u alignments, %u lanes x %d words]", todo[i].pl->max_m, todo[i].count,
                        todo[i].pl->members, todo[i].pl->sh.group, todo[i].pl->sh.cw);
            fprintf(stderr, "\n");
        }
        at = end;
    }
    if (!one_class) {
        uint64_t at = 0;
        for (uint32_t a = 0; a < n; a++) {
            c->h_offset[a] = at;
            at += cigar_len[a];
        }
        c->h_cigar.resize(at);
        for (uint32_t a = 0; a < n; a++)
            std::copy_n(stash.data() + stash_at[a], cigar_len[a], c->h_cigar.data() + c->h_offset[a]);
    }
    c->h_offset[n] = c->h_cigar.size();
    HIP_TRY(hipMemcpy(c->h_score.data(), c->out_score.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c->h_begin.data(), c->out_begin.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    *total_cigar = c->h_cigar.size();
    return BMV_OK;
}

int bmv_results(bmv_ctx *c, int32_t *out_score, uint32_t *out_begin, uint64_t *out_cigar_offset, uint32_t *out_cigar) {
    if (!c) return fail(BMV_ERR_ARG, "bmv_results: null context");
    if (out_score) memcpy(out_score, c->h_score.data(), c->h_score.size() * sizeof(int32_t));
    if (out_begin) memcpy(out_begin, c->h_begin.data(), c->h_begin.size() * sizeof(uint32_t));
    if (out_cigar_offset) memcpy(out_cigar_offset, c->h_offset.data(), c->h_offset.size() * sizeof(uint64_t));
    if (out_cigar && !c->h_cigar.empty()) memcpy(out_cigar, c->h_cigar.data(), c->h_cigar.size() * sizeof(uint32_t));
    return BMV_OK;
}

int bmv_last_stats(bmv_ctx *c, float *ms_kernels, uint64_t *n_cells) {
    if (!c) return fail(BMV_ERR_ARG, "bmv_last_stats: null context");
    if (ms_kernels) *ms_kernels = c->ms_kernels;
    if (n_cells) *n_cells = c->n_cells;
    return BMV_OK;
}

}  // extern "C"
