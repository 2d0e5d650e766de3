// bmf_api.hip -- C ABI (include/bmf.h) over the gfx950 kernels in bmf_kernels.hip.h.
//
// Host-side responsibilities only: parameter validation, HBM layout of the index
// (rows at a 128-byte pitch + one all-ones row), the fp64 sampler table (utils.h:160-178), launches
// and HIP-event timing.  There is deliberately no CPU implementation of the filter in this library.
#include "bmf_kernels.hip.h"
#include "bmf_vote2.hip.h"
#include "bmi_kernels.hip.h"
#include "bm_hip_util.h"
#include "bm_scan.hip.h"

#include "../../include/bmf.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
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

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(BMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                \
    } while (0)

// SeqAn3 dna4 assign_char folding; built once, uploaded per context.
void build_dna4_lut(uint8_t *lut) {
    memset(lut, 0, 256);
    const char *m[4] = {"AaRrWwMmDdHhVv", "CcYySsBb", "GgKk", "TtUu"};
    for (int r = 0; r < 4; r++)
        for (const char *c = m[r]; *c; c++) lut[(uint8_t)*c] = (uint8_t)r;
}

template <typename T>
hipError_t dev_alloc(T **p, size_t n) {
    return hipMalloc(reinterpret_cast<void **>(p), (n ? n : 1) * sizeof(T));
}

using vote_fn = void (*)(bmf::DevParams, const uint8_t *, const uint32_t *, const uint32_t *, uint32_t *,
                         uint32_t *, uint32_t *);

// rows in flight per wave: about 12-16 KB of row data per wave whatever the row length
constexpr int depth_for(int cpl) { return cpl <= 2 ? 8 : (cpl <= 4 ? 4 : (cpl <= 6 ? 3 : 2)); }

template <int CPL>
vote_fn pick_planes(int planes, bool prune) {
    constexpr int D = depth_for(CPL);
    switch (planes) {
    case 2: return prune ? bmf::bmf_vote_kernel<CPL, 2, D, false, true> : bmf::bmf_vote_kernel<CPL, 2, D, false, false>;
    case 3: return prune ? bmf::bmf_vote_kernel<CPL, 3, D, false, true> : bmf::bmf_vote_kernel<CPL, 3, D, false, false>;
    case 4: return prune ? bmf::bmf_vote_kernel<CPL, 4, D, false, true> : bmf::bmf_vote_kernel<CPL, 4, D, false, false>;
    case 5: return prune ? bmf::bmf_vote_kernel<CPL, 5, D, false, true> : bmf::bmf_vote_kernel<CPL, 5, D, false, false>;
    }
    return nullptr;
}

// NB > 65 536: every wave takes one slice of 8 chunks per lane
vote_fn pick_sliced(int planes, bool prune) {
    switch (planes) {
    case 2: return prune ? bmf::bmf_vote_kernel<8, 2, 2, true, true> : bmf::bmf_vote_kernel<8, 2, 2, true, false>;
    case 3: return prune ? bmf::bmf_vote_kernel<8, 3, 2, true, true> : bmf::bmf_vote_kernel<8, 3, 2, true, false>;
    case 4: return prune ? bmf::bmf_vote_kernel<8, 4, 2, true, true> : bmf::bmf_vote_kernel<8, 4, 2, true, false>;
    case 5: return prune ? bmf::bmf_vote_kernel<8, 5, 2, true, true> : bmf::bmf_vote_kernel<8, 5, 2, true, false>;
    }
    return nullptr;
}

vote_fn pick_vote(int cpl, int planes, bool prune) {
    switch (cpl) {
    case 1: return pick_planes<1>(planes, prune);
    case 2: return pick_planes<2>(planes, prune);
    case 3: return pick_planes<3>(planes, prune);
    case 4: return pick_planes<4>(planes, prune);
    case 5: return pick_planes<5>(planes, prune);
    case 6: return pick_planes<6>(planes, prune);
    case 7: return pick_planes<7>(planes, prune);
    case 8: return pick_planes<8>(planes, prune);
    }
    return nullptr;
}

// two-pass exact pruning (bmf_vote2.hip.h), unsliced geometries only
using pass1_fn = void (*)(bmf::DevParams, const uint8_t *, const uint32_t *, const uint32_t *, uint32_t *, bmf::Pass2Queue);
using recount_fn = void (*)(bmf::DevParams, const uint8_t *, const uint32_t *, uint32_t, uint32_t *, uint32_t *, bmf::Pass2Queue, uint32_t);
using finish_fn = void (*)(bmf::DevParams, const uint8_t *, const uint32_t *, uint32_t, uint32_t *, uint32_t *, bmf::Pass2Queue);
using slow_fn = void (*)(bmf::DevParams, const uint8_t *, const uint32_t *, uint32_t *, uint32_t *, bmf::Pass2Queue);

struct TwoPass {
    pass1_fn pass1 = nullptr;
    recount_fn recount = nullptr;
    slow_fn slow = nullptr;
    finish_fn finish = nullptr;
};

template <int CPL, int PLANES>
TwoPass two_pass_of(int max_live) {
    constexpr int D = depth_for(CPL);
    pass1_fn p1 = bmf::bmf_pass1_kernel<CPL, PLANES, D>;
    if constexpr (D > 2) {
        const char *env = getenv("BMF_PASS1_SHALLOW");
        if (env && env[0] == '1') p1 = bmf::bmf_pass1_kernel<CPL, PLANES, D - 1>;
    }
    recount_fn rc = max_live <= 16 ? bmf::bmf_recount_kernel<PLANES, 16> : bmf::bmf_recount_kernel<PLANES, 32>;
    return {p1, rc, bmf::bmf_vote2_slow_kernel<CPL, PLANES, D>, bmf::bmf_finish_kernel<PLANES>};
}

template <int CPL>
TwoPass pick_planes2(int planes, int max_live) {
    switch (planes) {
    case 2: return two_pass_of<CPL, 2>(max_live);
    case 3: return two_pass_of<CPL, 3>(max_live);
    case 4: return two_pass_of<CPL, 4>(max_live);
    case 5: return two_pass_of<CPL, 5>(max_live);
    }
    return {};
}

TwoPass pick_vote2(int cpl, int planes, int max_live) {
    switch (cpl) {
    case 1: return pick_planes2<1>(planes, max_live);
    case 2: return pick_planes2<2>(planes, max_live);
    case 3: return pick_planes2<3>(planes, max_live);
    case 4: return pick_planes2<4>(planes, max_live);
    case 5: return pick_planes2<5>(planes, max_live);
    case 6: return pick_planes2<6>(planes, max_live);
    case 7: return pick_planes2<7>(planes, max_live);
    case 8: return pick_planes2<8>(planes, max_live);
    }
    return {};
}

// first pass over a folded index: NB <= 65 536 folds by 4 to at most 128 chunks per row (CPL 1, 2), by 2 to 256 (1..4)
template <int CPL, int FOLD>
pass1_fn pick_fold_planes(int planes) {
    constexpr int D = depth_for(CPL);
    switch (planes) {
    case 2: return bmf::bmf_pass1_kernel<CPL, 2, D, FOLD>;
    case 3: return bmf::bmf_pass1_kernel<CPL, 3, D, FOLD>;
    case 4: return bmf::bmf_pass1_kernel<CPL, 4, D, FOLD>;
    case 5: return bmf::bmf_pass1_kernel<CPL, 5, D, FOLD>;
    }
    return nullptr;
}
pass1_fn pick_pass1_fold(int fold, int cpl, int planes) {
    if (fold == 4) return cpl == 1 ? pick_fold_planes<1, 4>(planes) : (cpl == 2 ? pick_fold_planes<2, 4>(planes) : nullptr);
    if (fold == 2) {
        switch (cpl) {
        case 1: return pick_fold_planes<1, 2>(planes);
        case 2: return pick_fold_planes<2, 2>(planes);
        case 3: return pick_fold_planes<3, 2>(planes);
        case 4: return pick_fold_planes<4, 2>(planes);
        }
    }
    return nullptr;
}

// P[Bin(n, p) >= m]
double binom_tail(uint32_t n, double p, uint32_t m) {
    if (m == 0) return 1.0;
    if (m > n) return 0.0;
    double tail = 0.0;
    for (uint32_t i = m; i <= n; i++) {
        double c = 1.0;
        for (uint32_t j = 0; j < i; j++) c = c * (double)(n - j) / (double)(j + 1);
        tail += c * pow(p, (double)i) * pow(1.0 - p, (double)(n - i));
    }
    return tail;
}

}  // namespace

// Grow-only device buffer (a batch that is reused keeps its allocations).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n) {
        if (n <= cap && p) return hipSuccess;
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

struct bmf_batch {
    uint32_t n_windows = 0;
    uint64_t n_bytes = 0;
    DevBuf<uint8_t> bases, quals, scan_tmp;
    DevBuf<uint64_t> win_start;
    DevBuf<uint32_t> win_len, lists, list_n, rows_anded, counts, buckets, offsets, compact;
    DevBuf<uint32_t> slice_min, slice_cnt, slice_ids;   // NB > 65 536 only
    DevBuf<uint32_t> q_counters, q_slow, q_live_n, q_left;   // two-pass pruning only (bmf::Pass2Queue)
    DevBuf<uint16_t> q_live_chunks;
    DevBuf<uint4> q_live_mask;
};

struct bmf_ctx {
    bmf_params p{};
    bmf::DevParams dp{};
    hipStream_t stream = nullptr;
    // index in HBM
    bool loaded = false;
    uint64_t n_rows = 0;
    uint8_t *d_rows = nullptr;       // (n_rows + 1) x pitch ; row n_rows is all ones
    int32_t *d_k2i = nullptr;
    uint32_t *d_zeros = nullptr;
    uint32_t *d_qgram_ok = nullptr;  // bitmap over 4^q q-grams
    // constants
    uint8_t *d_lut = nullptr;
    uint16_t *d_pos_table = nullptr;
    // kernel variant
    int cpl = 0, planes = 0, depth = 0;
    uint32_t n_slices = 1;           // > 1 when NB > 65 536: one wave per (item, 65 536-bucket slice)
    vote_fn vote = nullptr;
    TwoPass two_pass;                // used instead of `vote` when dp.pass1_rows > 0
    // folded first pass (bmf_fold4_kernel): one bit per group of 4 buckets, a quarter of the row bytes
    uint8_t *d_fold = nullptr;       // (n_rows + 1) x dpf.pitch
    DevBuf<uint32_t> tune_lists;     // tune_pruned: the measured prefix's row-id lists as the sample kernel wrote them
    bmf::DevParams dpf{};            // the folded geometry + rows per sample the folded pass reads
    uint32_t fold = 1;               // 2 or 4 when pass1_fold is set
    // The folded pass is chosen by a model of how many chunks survive it by chance.  Guard: every run's slow-path
    // count comes back asynchronously; if more than 2 % of a run's items overflowed the recount kernel's lanes the
    // model was wrong for this index and the context falls back to the unfolded choice for the runs that follow
    // (outputs are identical either way).
    uint32_t unfolded_rows = 0;      // pass 1 rows of the unfolded choice (0: the single-pass pruning kernel)
    uint32_t *h_guard = nullptr;     // pinned: [recounted, slow, loads, -] of the last folded run
    hipEvent_t guard_ev = nullptr;
    uint64_t guard_items = 0;
    bool guard_pending = false;
    pass1_fn pass1_fold = nullptr;   // non-null: pass 1 streams d_fold instead of d_rows
    int unfolded_max_live = 32;      // lanes per item of the recount kernel that go with unfolded_rows
    bool sort_rows = true;           // two-pass forms: each sample's rows sparsest first (BMF_ROW_ORDER=far|linear: not)
    bool tunable = false, tuned = false;   // the pruning form is measured on the first large batch (tune_pruned)
    uint32_t tune_windows = 32768;
    double guard_baseline = -1.0;    // share of slow-path items of the first run after tuning (< 0: not sampled yet)
    unsigned recount_waves = 4096;   // waves of the recount kernel the device holds at once (CUs x 4 SIMDs x BMF_RECOUNT_OCC)
    bool no_finish = false;          // BMF_NO_FINISH=1: every item through the recount kernel (experiments)
    size_t sample_lds = 0;
    bmf::SampleGeom sample_geom{};
    bool sample_bitmap_lds = false;
    // profiling
    bool profiling = false;
    uint32_t prof_max = 0, prof_n = 0;
    std::vector<hipEvent_t> ev;      // 3 per run: before sample, between, after vote
    // bmf_map_windows: a batch is cut into pieces; while piece i runs on `stream`, the next pieces are uploaded on
    // `h2d` and the results of the previous one come back on `d2h`.  Three slots of device + pinned staging
    // buffers: two pieces are queued behind the one whose results the host is unpacking.
    struct MapSlot {
        bmf_batch dev;
        DevBuf<uint32_t> pack;           // [total | ids of all lists back to back]
        uint8_t *h_views = nullptr;      // pinned: rebased win_start (u64 x n) then win_len (u32 x n)
        size_t h_views_cap = 0;
        uint32_t *h_out = nullptr;       // pinned: counts (2n) then the head of `pack`
        size_t h_out_cap = 0;
        uint8_t *h_bases = nullptr, *h_quals = nullptr;   // pinned: the piece's windows gathered back to back (bmf_map_text_windows_compact)
        size_t h_bases_cap = 0, h_quals_cap = 0;
        hipEvent_t uploaded = nullptr, ran = nullptr, landed = nullptr;
        // the piece in flight
        uint32_t first = 0, n = 0;
        size_t ids_copied = 0;
    };
    static constexpr int kMapSlots = 3;
    MapSlot *slot[kMapSlots] = {nullptr, nullptr, nullptr};
    hipStream_t h2d = nullptr, d2h = nullptr;
    // two-pass pruning: the batch goes out in slices, the recount of slice i on `side` under pass 1 of slice i+1
    static constexpr int kSlices = 8;
    hipStream_t side = nullptr;
    hipEvent_t slice_done[kSlices] = {};
    hipEvent_t side_done = nullptr;
    DevBuf<uint8_t> whole_bases, whole_quals;   // windows in no particular order: the read buffer is uploaded once
};

extern "C" {

int bmf_abi_version(void) { return BMF_ABI_VERSION; }
const char *bmf_last_error(void) { return g_err; }

// main.cpp:207 -- unsigned * float is a float32 product, then ceil
uint32_t bmf_fault_from_rate(uint32_t samples, float max_error_rate) {
    volatile float prod = (float)samples * max_error_rate;
    return (uint32_t)ceil((double)prod);
}
// q_gram_mapper.h:163
uint32_t bmf_threshold(float distinguishability, uint32_t num_buckets) {
    volatile float prod = distinguishability * (float)num_buckets;
    return (uint32_t)prod;
}
// q_gram_mapper.h:510-516 (Sampler: utils.h:160-178)
uint32_t bmf_window_starts(uint32_t record_len, uint32_t read_len, uint32_t n_seg, uint32_t *out) {
    if ((uint64_t)record_len > 2ull * read_len && n_seg > 0) {
        const uint32_t ub = record_len - read_len - 1u;
        double delta = 0.0;
        if (n_seg != 1) delta = (double)(ub + 1u) / (double)(n_seg - 1u);
        for (uint32_t i = 0; i + 1 < n_seg; i++) out[i] = (uint32_t)floor((double)i * delta);
        out[n_seg - 1] = ub;
        return n_seg;
    }
    out[0] = 0;
    return 1;
}
// bucket_locator.h:419-420
uint32_t bmf_ceil_mul_f32(float rate, uint32_t n) {
    volatile float prod = rate * (float)n;
    return (uint32_t)ceil((double)prod);
}

int bmf_create(const bmf_params *params, bmf_ctx **out) {
    if (!params || !out) return fail(BMF_ERR_ARG, "bmf_create: null argument");
    *out = nullptr;
    const bmf_params &p = *params;
    if (p.num_buckets == 0) return fail(BMF_ERR_ARG, "num_buckets must be > 0");
    if (p.q == 0 || p.q > 15 || p.k < p.q || p.k > 16)
        return fail(BMF_ERR_ARG, "need 1 <= q <= 15 and q <= k <= 16 (got q=%u k=%u)", p.q, p.k);
    if (p.num_samples == 0 || p.num_samples > 64)
        return fail(BMF_ERR_UNSUPPORTED, "num_samples must be in 1..64 (got %u)", p.num_samples);
    if (p.num_fault == 0 || p.num_fault > 31)
        return fail(BMF_ERR_UNSUPPORTED, "num_fault must be in 1..31 (got %u)", p.num_fault);
    if (p.max_candidates == 0 || p.max_candidates > 64)
        return fail(BMF_ERR_UNSUPPORTED, "max_candidates must be in 1..64 (got %u)", p.max_candidates);
    if (p.read_len < p.k || p.read_len > 16384)
        return fail(BMF_ERR_UNSUPPORTED, "read_len must be in k..16384 (got %u)", p.read_len);
    if (p.k - p.q + 1 > 8) return fail(BMF_ERR_UNSUPPORTED, "k-q+1 must be <= 8");

    const uint32_t row_bytes = (p.num_buckets + 7u) >> 3;
    const uint32_t n_chunks = (row_bytes + 15u) / 16u;
    int cpl = (int)((n_chunks + 63u) / 64u);
    const uint32_t n_slices = cpl > 8 ? (n_chunks + 511u) / 512u : 1u;   // NB > 65 536: 65 536-bucket slices
    if (n_slices > 256) return fail(BMF_ERR_UNSUPPORTED, "num_buckets must be <= 16777216 (got %u)", p.num_buckets);
    if (n_slices > 1) cpl = 8;
    int planes = 0;
    while (((1u << planes) - 1u) < p.num_fault) planes++;
    if (planes < 2) planes = 2;
    if (const char *env = getenv("BMF_MIN_PLANES")) {   // experiments: a wider counter than F needs (same outputs)
        const int v = atoi(env);
        if (v >= 2 && v <= 5 && v > planes) planes = v;
    }

    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (p.device < 0 || p.device >= n_dev)
        return fail(BMF_ERR_HIP, "device %d not available (%d HIP devices)", p.device, n_dev);
    HIP_TRY(hipSetDevice(p.device));

    bmf_ctx *c = new bmf_ctx();
    c->p = p;
    c->cpl = cpl;
    c->planes = planes;
    c->n_slices = n_slices;
    c->depth = n_slices > 1 ? 2 : depth_for(cpl);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p.device) == hipSuccess && cus > 0)
            c->recount_waves = (unsigned)cus * 4u * BMF_RECOUNT_OCC;
        if (const char *env = getenv("BMF_RECOUNT_WAVES")) c->recount_waves = (unsigned)std::max(64, atoi(env));   // experiments
        c->no_finish = getenv("BMF_NO_FINISH") != nullptr;
    }
    const bool prune = (p.flags & BMF_FLAG_EARLY_EXIT) != 0;
    c->vote = n_slices > 1 ? pick_sliced(planes, prune) : pick_vote(cpl, planes, prune);
    if (!c->vote) {
        delete c;
        return fail(BMF_ERR_UNSUPPORTED, "no vote kernel for cpl=%d planes=%d", cpl, planes);
    }
    bmf::DevParams &d = c->dp;
    d.nb = p.num_buckets;
    d.k = p.k;
    d.q = p.q;
    d.G = p.k - p.q + 1;
    d.S = p.num_samples;
    d.F = p.num_fault;
    d.qbits = (uint32_t)((1ull << (2 * p.q)) - 1ull);
    d.minq = p.min_base_quality;
    // q_gram_mapper.h:445: reject iff (double)size < 0.2 * num_samples
    d.min_good = 0;
    while ((double)d.min_good < 0.2 * (double)p.num_samples) d.min_good++;
    // (size()-1 underflows in the reference when nothing is good and S == 0 only; S >= 1 here, but
    //  n_good == 0 must never reach the sampler.)
    if (d.min_good == 0) d.min_good = 1;
    d.max_cand = p.max_candidates;
    d.read_len = p.read_len;
    d.max_kmers = p.read_len - p.k + 1;
    // S*G row ids, rounded up to the ring depth, plus one ring of padding (all-ones rows)
    d.list_len = (d.S * d.G + (uint32_t)c->depth - 1u) / (uint32_t)c->depth * (uint32_t)c->depth + (uint32_t)c->depth;
    d.n_chunks = n_chunks;
    d.pitch = (row_bytes + 127u) & ~127u;
    d.ones_row = 0;
    d.n_kmers = 0;
    d.early_exit = (p.flags & BMF_FLAG_EARLY_EXIT) ? 1u : 0u;
    {   // Order of a sample's G row ids in the lists the sample kernel writes: farthest-point order of the q-gram
        // numbers -- 0, G-1, then whatever lies farthest from those taken (bmf_vote2.hip.h says why).  A permutation:
        // the vote kernel ANDs all G and does not care.  BMF_ROW_ORDER=linear keeps 0, 1, 2, ... for comparison.
        uint32_t order[8] = {0, 1, 2, 3, 4, 5, 6, 7}, n = 0;
        const bool linear = getenv("BMF_ROW_ORDER") && !strcmp(getenv("BMF_ROW_ORDER"), "linear");
        bool taken[8] = {};
        while (!linear && n < d.G) {
            uint32_t pick = 0;
            int best_dist = -1;
            for (uint32_t g = 0; g < d.G; g++) {
                if (taken[g]) continue;
                int dist = 99;
                for (uint32_t t = 0; t < n; t++) dist = std::min(dist, abs((int)g - (int)order[t]));
                if (n == 0) dist = g == 0 ? 99 : 0;
                if (dist > best_dist || (dist == best_dist && g > pick)) {
                    best_dist = dist;
                    pick = g;
                }
            }
            taken[pick] = true;
            order[n++] = pick;
        }
        d.row_order = 0;
        for (uint32_t i = 0; i < d.G; i++) d.row_order |= order[i] << (4u * i);
    }
    // Sample kernel geometry: waves (= windows in flight) per workgroup and whether the 4^q-bit q-gram bitmap is
    // staged in LDS.  Two workgroups per CU (80 KiB each) where the buffers allow it: 32 waves per CU.
    {
        bmf::SampleGeom &g = c->sample_geom;
        g.bitmap_words = (uint32_t)(((1ull << (2 * p.q)) + 31) / 32);
        const uint32_t stream = (p.read_len + 14u + 15u) & ~15u;   // stream positions of the aligned 8-byte chunks covering a window
        g.pk_bytes = ((stream / 16u + 3u) * 4u + 15u) & ~15u;      // + the words the last positions' shifts touch
        g.qsum_bytes = ((stream + 1u) * 4u + 15u) & ~15u;
        g.wave_stride = (g.pk_bytes + g.qsum_bytes + 4u * d.max_kmers + 15u) & ~15u;
        const size_t bitmap_bytes = ((size_t)g.bitmap_words * 4 + 15) & ~(size_t)15;
        const size_t half = 80 * 1024, full = 160 * 1024 - 1024;
        auto waves_in = [&](size_t budget, size_t fixed) -> uint32_t {
            return budget > fixed ? (uint32_t)std::min<size_t>(16, (budget - fixed) / g.wave_stride) : 0u;
        };
        c->sample_bitmap_lds = true;
        g.waves_per_wg = waves_in(half, bitmap_bytes);
        if (g.waves_per_wg < 8) g.waves_per_wg = waves_in(full, bitmap_bytes);
        if (g.waves_per_wg < 1) {   // the bitmap does not fit beside even one window: it stays in L2
            c->sample_bitmap_lds = false;
            g.waves_per_wg = waves_in(half, 0);
            if (g.waves_per_wg < 4) g.waves_per_wg = waves_in(full, 0);
        }
        if (g.waves_per_wg < 1) {
            delete c;
            return fail(BMF_ERR_UNSUPPORTED, "read_len %u does not fit the sample kernel's LDS buffers", p.read_len);
        }
        c->sample_lds = (c->sample_bitmap_lds ? bitmap_bytes : 0) + (size_t)g.waves_per_wg * g.wave_stride;
    }

    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(BMF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    // Sampler::sample_deterministically (utils.h:160-178) tabulated for every possible number of
    // good k-mers n (upper_bound = n-1), in fp64 on the host so device rounding can never differ.
    // upper_bound == 0: the reference skips re-sampling and reads out of bounds; we define it as
    // S zeros (see DESIGN.md "deviations").
    std::vector<uint16_t> tab((size_t)(d.max_kmers + 1) * d.S, 0);
    for (uint32_t n = 1; n <= d.max_kmers; n++) {
        const uint32_t ub = n - 1;
        double delta = 0.0;
        if (d.S != 1) delta = (double)(ub + 1u) / (double)(d.S - 1u);
        for (uint32_t i = 0; i + 1 < d.S; i++) tab[(size_t)n * d.S + i] = (uint16_t)floor((double)i * delta);
        tab[(size_t)n * d.S + d.S - 1] = (uint16_t)ub;
    }
    uint8_t lut[256];
    build_dna4_lut(lut);
    if (dev_alloc(&c->d_pos_table, tab.size()) != hipSuccess || dev_alloc(&c->d_lut, 256) != hipSuccess ||
        hipMemcpy(c->d_pos_table, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_lut, lut, 256, hipMemcpyHostToDevice) != hipSuccess) {
        bmf_destroy(c);
        return fail(BMF_ERR_HIP, "uploading sampler table failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (c->sample_lds > 48 * 1024) {
        e = bmhip::raise_dynamic_lds(c->sample_bitmap_lds ? reinterpret_cast<const void *>(bmf::bmf_sample_kernel<true>)
                                                   : reinterpret_cast<const void *>(bmf::bmf_sample_kernel<false>), c->sample_lds);
        if (e != hipSuccess) {
            bmf_destroy(c);
            return fail(BMF_ERR_HIP, "cannot reserve %zu B of LDS: %s", c->sample_lds, hipGetErrorString(e));
        }
    }
    *out = c;
    return BMF_OK;
}

static void free_index(bmf_ctx *c) {
    (void)hipFree(c->d_rows);
    (void)hipFree(c->d_k2i);
    (void)hipFree(c->d_zeros);
    (void)hipFree(c->d_qgram_ok);
    (void)hipFree(c->d_fold);
    c->d_fold = nullptr;
    c->pass1_fold = nullptr;
    c->guard_pending = false;
    c->d_rows = nullptr;
    c->d_k2i = nullptr;
    c->d_zeros = nullptr;
    c->d_qgram_ok = nullptr;
    c->n_rows = 0;
    c->loaded = false;
}

// The folded copy of the index for a first pass that reads `fold_r` rows per sample of one bit per `fold_f` buckets:
// sets c->dpf / c->pass1_fold / c->fold and (re)builds c->d_fold when the fold factor changes.
static int build_fold(bmf_ctx *c, uint32_t fold_f, uint32_t fold_r) {
    const bmf::DevParams &d = c->dp;
    bmf::DevParams &f = c->dpf;
    f = c->dp;
    f.nb = (d.nb + fold_f - 1u) / fold_f;
    const uint32_t row_bytes_f = (f.nb + 7u) >> 3;
    f.n_chunks = (row_bytes_f + 15u) / 16u;
    f.pitch = (row_bytes_f + 127u) & ~127u;
    f.pass1_rows = fold_r;
    const int cpl_f = (int)((f.n_chunks + 63u) / 64u);
    c->pass1_fold = pick_pass1_fold((int)fold_f, cpl_f, c->planes);
    if (!c->pass1_fold) return BMF_OK;
    if (c->d_fold && c->fold == fold_f) return BMF_OK;             // the copy is there already
    if (c->d_fold) (void)hipFree(c->d_fold);
    c->d_fold = nullptr;
    c->fold = fold_f;
    HIP_TRY(dev_alloc(&c->d_fold, (size_t)(c->n_rows + 1) * f.pitch));
    // whole rows per launch, grid.x * 256 threads below 2^32
    const uint64_t rows_per_launch = std::max<uint64_t>(1, ((uint64_t)1 << 30) / (f.pitch >> 2));
    for (uint64_t r0 = 0; r0 <= c->n_rows; r0 += rows_per_launch) {
        const uint64_t nr = std::min<uint64_t>(rows_per_launch, c->n_rows + 1 - r0);
        const uint64_t w = nr * (f.pitch >> 2);
        auto fk = fold_f == 4 ? bmf::bmf_fold_kernel<4> : bmf::bmf_fold_kernel<2>;
        hipLaunchKernelGGL(fk, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_rows + (size_t)r0 * c->dp.pitch, nr, c->dp.pitch, c->d_fold + (size_t)r0 * f.pitch, f.pitch);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BMF_OK;
}


// BMF_FLAG_EARLY_EXIT: which exact-pruning kernel serves this index.  The two-pass kernel streams r rows
// per sample at full width, then recounts the chunks that survive; a bucket unrelated to the read survives
// a sample with probability about h^r, h = 1 - f + f*d (d = density of the rows a read meets, weighted by density
// because a q-gram is met in proportion to how often it occurs; f = fraction of q-grams FracMinHash kept -- the
// others AND as the identity), so the expected number of survivors is NB * P[Bin(S, h^r) >= S-F+1].  Costs are in row bytes; the single-pass PRUNE kernel reads F*G whole rows
// before it can narrow.  BMF_PASS1_ROWS=r forces r (0: never two-pass) for experiments.
static int select_pruned_variant(bmf_ctx *c) {
    const bmf::DevParams &d = c->dp;
    c->sort_rows = !getenv("BMF_ROW_ORDER");
    c->tunable = c->tuned = false;
    c->guard_baseline = -1.0;
    c->dp.pass1_rows = 0;
    c->dp.max_live = bmf::kMaxLive;
    c->dp.item_base = 0;
    if (!(c->p.flags & BMF_FLAG_EARLY_EXIT) || c->n_slices > 1) return BMF_OK;
    c->vote = pick_vote(c->cpl, c->planes, true);
    if (d.G < 2 || c->n_rows == 0) return BMF_OK;
    std::vector<uint32_t> zeros((size_t)c->n_rows);
    HIP_TRY(hipMemcpy(zeros.data(), c->d_zeros, zeros.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    double s1 = 0.0, s2 = 0.0;
    for (uint32_t z : zeros) {
        const double di = 1.0 - (double)z / (double)d.nb;
        s1 += di;
        s2 += di * di;
    }
    const double dens = s1 > 0.0 ? s2 / s1 : 0.0;
    // a q-gram FracMinHash did not keep ANDs as the identity: a row of a sample tells nothing with probability 1 - f
    const double kept = d.n_kmers ? (double)c->n_rows / (double)d.n_kmers : 1.0;
    const double hit1 = 1.0 - kept + kept * dens;   // P[one row of a sample leaves an unrelated bucket's bit set]
    // With most q-grams not indexed (the reference's default -f 0.25) three rows in four are the cache-resident
    // all-ones row: the plain kernel then runs at 50+ M reads/s and the pruning kernels' bookkeeping costs more
    // than the few real rows they skip (measured: 53.6 M plain, 32.0 M single-pass pruning on the Egu -f 0.25
    // index).  The flag promises identical outputs from no more work, so the plain kernel serves such an index.
    // The same holds for very short rows (NB <= 2 048): the step is latency-bound and the plain kernel is the
    // fastest (E. coli-sized index: 177 M reads/s plain, 160 M with single-pass pruning; from NB ~ 3 800 on,
    // pruning wins again: 102 M plain, 129 M single-pass).
    const bool fold_forced = getenv("BMF_FOLD") && atoi(getenv("BMF_FOLD")) > 1;
    if ((kept < 0.6 || d.n_chunks <= 16u) && !getenv("BMF_PASS1_ROWS") && !fold_forced) {
        c->vote = pick_vote(c->cpl, c->planes, false);
        return BMF_OK;
    }
    {   // from here on a pruning form serves the index: which one is measured on the first large batch (tune_pruned)
        const char *at = getenv("BMF_AUTOTUNE");
        const bool forced = getenv("BMF_PASS1_ROWS") || getenv("BMF_FOLD") || getenv("BMF_MAX_LIVE") || getenv("BMF_ROW_ORDER") ||
                            getenv("BMF_SLICES") || getenv("BMF_GUARD_TRIP");
        c->tunable = !forced && !(at && at[0] == '0');
        c->tune_windows = 32768;
        if (const char *tw = getenv("BMF_TUNE_WINDOWS")) c->tune_windows = (uint32_t)std::max(64l, atol(tw));
    }
    const double row_bytes = (double)d.n_chunks * 16.0, sector = 64.0;
    const double prune_cost = (double)d.F * d.G * row_bytes;
    uint32_t best_r = 0;
    double best = 0.95 * prune_cost, best_live = 0.0;
    for (uint32_t r = 1; r < d.G; r++) {
        const double live = (double)d.nb * binom_tail(d.S, pow(hit1, (double)r), d.S - d.F + 1u);
        if (live > 20.0) continue;   // more than kMaxLive live chunks send an item down the slow path
        const double cost = (double)d.S * r * row_bytes + live * d.S * sector + 1.0 * d.S * d.G * sector;
        if (cost < best) {
            best = cost;
            best_r = r;
            best_live = live;
        }
    }
    if (const char *env = getenv("BMF_PASS1_ROWS")) {
        const long v = strtol(env, nullptr, 10);
        best_r = v > 0 && (uint32_t)v < d.G ? (uint32_t)v : 0u;
        if (best_r) best_live = (double)d.nb * binom_tail(d.S, pow(hit1, (double)best_r), d.S - d.F + 1u);
    }
    // The folded first pass: a row of the index folded by f (one bit per group of f buckets) is 1/f of the bytes and
    // leaves a group's bit set with probability 1 - (1 - d)^f, so r rows of it per sample cost what r/f rows cost
    // now.  With the rows taken far apart (bmf_vote2.hip.h) they are close to independent and r of them let an
    // unrelated group through with probability ~ (1 - (1 - d)^f)^r: at d = 0.22, two half-width rows cost what one
    // full row costs and leave 0.1 instead of 7.6 chunks alive by chance.  Rows closer than 3 q-gram positions are
    // NOT independent (an occurrence of one q-gram continues into the next with probability 4^-shift); the model
    // prices that in, roughly.  Same cost model as above, groups instead of buckets.  BMF_FOLD=0|2|4 and
    // BMF_FOLD_ROWS=r override (the sweeps force every form).
    uint32_t fold_f = 1, fold_r = 0;
    double fold_live = 0.0;
    {
        uint32_t order[8];
        for (uint32_t i = 0; i < d.G; i++) order[i] = (d.row_order >> (4u * i)) & 15u;
        auto survivors = [&](uint32_t f, uint32_t r) {
            const double lambda = -log(1.0 - dens) * f, df = 1.0 - exp(-lambda);      // density of a folded row
            const double occ = lambda / std::max(1e-9, df);                             // occurrences of a q-gram in a group that holds it
            double p = 1.0;
            for (uint32_t i = 0; i < r; i++) {
                int shift = 99;
                for (uint32_t t = 0; t < i; t++) shift = std::min(shift, abs((int)order[i] - (int)order[t]));
                const double cont = i ? 1.0 - pow(1.0 - pow(0.25, (double)shift), occ) : 0.0;   // P[a neighbour's hit continues into this q-gram]
                const double di = df + (1.0 - df) * cont;
                p *= 1.0 - kept + kept * di;
            }
            return (((double)d.nb + f - 1.0) / f) * binom_tail(d.S, p, d.S - d.F + 1u);
        };
        double best_cost = best;
        for (uint32_t f : {2u, 4u}) {
            const uint32_t row_bytes_f = (uint32_t)((((double)d.nb + f - 1.0) / f + 7.0) / 8.0), chunks_f = (row_bytes_f + 15u) / 16u;
            if (chunks_f > (f == 2 ? 256u : 128u)) continue;
            for (uint32_t r = 1; r <= d.G; r++) {
                const double live = survivors(f, r);
                if (live > 8.0) continue;            // the model is rough: stay well clear of the slow path
                const double cost = (double)d.S * r * (row_bytes / f) + live * d.S * sector + 1.0 * d.S * d.G * sector;
                if (cost < best_cost) {
                    best_cost = cost;
                    fold_f = f;
                    fold_r = r;
                    fold_live = live;
                }
            }
        }
        const char *ef = getenv("BMF_FOLD");
        const int forced = ef ? atoi(ef) : -1;
        if (forced == 0 || (getenv("BMF_PASS1_ROWS") && forced <= 0)) fold_r = 0;   // an experiment asked for the unfolded passes
        if (forced == 2 || forced == 4) {
            const long v = getenv("BMF_FOLD_ROWS") ? strtol(getenv("BMF_FOLD_ROWS"), nullptr, 10) : ((uint32_t)forced == fold_f && fold_r ? (long)fold_r : (long)d.G);
            fold_f = (uint32_t)forced;
            fold_r = (uint32_t)std::max<long>(1, std::min<long>(v, (long)d.G));
            fold_live = survivors(fold_f, fold_r);
        }
        c->unfolded_rows = best_r;
        // the recount kernel's form that goes with the UNFOLDED choice (the folded model keeps fold_live <= 8 and so
        // always asks for 16 lanes; the unfolded passes may leave ~20 chunks alive by chance): the guard restores both
        c->unfolded_max_live = best_live + 1.0 <= 9.0 ? 16 : 32;
        if (fold_r) {
            best_r = best_r ? best_r : 1u;               // what the recount and the slow kernel call "pass 1's rows"
            best_live = fold_live;
        }
    }
    if (best_r) {
        // Lanes per item in the recount kernel: 16 (four items per wave) while items with more than 16 live chunks stay
        // rare -- the by-chance survivors are Poisson around best_live, plus the read's own chunk -- else 32.
        // BMF_MAX_LIVE=16|32 overrides (the sweeps force both).
        int max_live = best_live + 1.0 <= 9.0 ? 16 : 32;
        if (const char *env = getenv("BMF_MAX_LIVE")) max_live = atoi(env) == 16 ? 16 : 32;
        c->two_pass = pick_vote2(c->cpl, c->planes, max_live);
        if (c->two_pass.pass1) {
            c->dp.pass1_rows = best_r;
            c->dp.max_live = (uint32_t)max_live;
        }
    }
    if (fold_r && c->dp.pass1_rows) return build_fold(c, fold_f, fold_r);
    return BMF_OK;
}

static void release_batch(bmf_batch *b) {
    b->bases.release(); b->quals.release(); b->scan_tmp.release(); b->win_start.release(); b->win_len.release();
    b->lists.release(); b->list_n.release(); b->rows_anded.release(); b->counts.release(); b->buckets.release();
    b->offsets.release(); b->compact.release();
    b->slice_min.release(); b->slice_cnt.release(); b->slice_ids.release();
    b->q_counters.release(); b->q_slow.release(); b->q_live_n.release(); b->q_left.release(); b->q_live_chunks.release(); b->q_live_mask.release();
}

static void free_map_slots(bmf_ctx *c) {
    for (auto *&sl : c->slot) {
        if (!sl) continue;
        release_batch(&sl->dev);
        sl->pack.release();
        if (sl->h_views) (void)hipHostFree(sl->h_views);
        if (sl->h_out) (void)hipHostFree(sl->h_out);
        if (sl->h_bases) (void)hipHostFree(sl->h_bases);
        if (sl->h_quals) (void)hipHostFree(sl->h_quals);
        for (hipEvent_t e : {sl->uploaded, sl->ran, sl->landed})
            if (e) (void)hipEventDestroy(e);
        delete sl;
        sl = nullptr;
    }
    c->whole_bases.release();
    c->whole_quals.release();
    if (c->h2d) (void)hipStreamDestroy(c->h2d);
    if (c->d2h) (void)hipStreamDestroy(c->d2h);
    c->h2d = c->d2h = nullptr;
    if (c->side) (void)hipStreamDestroy(c->side);
    c->side = nullptr;
    if (c->h_guard) (void)hipHostFree(c->h_guard);
    c->h_guard = nullptr;
    if (c->guard_ev) (void)hipEventDestroy(c->guard_ev);
    c->guard_ev = nullptr;
    c->guard_pending = false;
    for (auto &e : c->slice_done) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (c->side_done) (void)hipEventDestroy(c->side_done);
    c->side_done = nullptr;
}

void bmf_destroy(bmf_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    free_map_slots(c);
    free_index(c);
    c->tune_lists.release();
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_pos_table);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// The three steps of an index upload, shared by bmf_load_index (rows in memory) and bmf_load_index_files
// (rows streamed from the .qgram file through pinned staging buffers).
static int upload_begin(bmf_ctx *c, uint64_t n_rows, const int32_t *kmer_to_index, uint64_t n_kmers) {
    // q_gram_mapper.h:325-328: "The q-gram index is not empty. Terminating load."
    if (c->loaded) return fail(BMF_ERR_STATE, "the q-gram index is not empty; call bmf_reset first");
    if (n_kmers && !kmer_to_index) return fail(BMF_ERR_ARG, "kmer_to_index is null");
    if (n_kmers != 0 && n_kmers != (1ull << (2 * c->p.q)))
        return fail(BMF_ERR_ARG, "kmer_to_index must have 4^q = %llu entries (got %llu)",
                    (unsigned long long)(1ull << (2 * c->p.q)), (unsigned long long)n_kmers);
    if (n_rows >= 0x7FFFFFFFull) return fail(BMF_ERR_ARG, "too many rows");
    for (uint64_t i = 0; i < n_kmers; i++)
        if (kmer_to_index[i] >= 0 && (uint64_t)kmer_to_index[i] >= n_rows)
            return fail(BMF_ERR_ARG, "kmer_to_index[%llu] = %d is not a row (n_rows = %llu)", (unsigned long long)i,
                        kmer_to_index[i], (unsigned long long)n_rows);
    HIP_TRY(hipSetDevice(c->p.device));
    const uint32_t pitch = c->dp.pitch;
    const uint64_t n_words = ((1ull << (2 * c->p.q)) + 31) / 32;
    HIP_TRY(dev_alloc(&c->d_rows, (size_t)(n_rows + 1) * pitch));
    HIP_TRY(dev_alloc(&c->d_k2i, (size_t)n_kmers));
    HIP_TRY(dev_alloc(&c->d_zeros, (size_t)n_rows));
    HIP_TRY(dev_alloc(&c->d_qgram_ok, (size_t)n_words));
    HIP_TRY(hipMemsetAsync(c->d_rows, 0, (size_t)n_rows * pitch, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_rows + (size_t)n_rows * pitch, 0xFF, pitch, c->stream));
    if (n_kmers) HIP_TRY(hipMemcpyAsync(c->d_k2i, kmer_to_index, (size_t)n_kmers * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_rows = n_rows;
    c->dp.ones_row = (uint32_t)n_rows;
    c->dp.n_kmers = (uint32_t)n_kmers;
    return BMF_OK;
}

// rows [first, first + n) in the .qgram layout, to their 128-byte-pitched slots: one flat copy into a device
// staging buffer, then bmf_repitch_kernel (a strided 2-D copy from the host runs at ~0.45 GB/s here)
static hipError_t upload_rows(bmf_ctx *c, uint64_t first, uint64_t n, const uint8_t *rows, uint8_t *d_stage) {
    const uint32_t pitch = c->dp.pitch, row_bytes = (c->p.num_buckets + 7u) >> 3;
    hipError_t e = hipMemcpyAsync(d_stage, rows, (size_t)n * row_bytes, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return e;
    const uint64_t work = n * ((row_bytes + 3u) / 4u);
    hipLaunchKernelGGL(bmf::bmf_repitch_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, c->stream, d_stage,
                       c->d_rows + (size_t)first * pitch, n, row_bytes, pitch);
    return hipGetLastError();
}

// rows per piece of an upload (32 MiB)
static uint64_t piece_rows_of(uint32_t row_bytes) { return std::max<uint64_t>(1, ((uint64_t)32 << 20) / std::max(1u, row_bytes)); }

static int upload_finish(bmf_ctx *c) {
    const uint32_t pitch = c->dp.pitch;
    const uint64_t n_rows = c->n_rows, n_kmers = c->dp.n_kmers;
    const uint64_t n_words = ((1ull << (2 * c->p.q)) + 31) / 32;
    if (n_rows) {
        hipLaunchKernelGGL(bmf::bmf_sanitize_rows_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_rows, n_rows, pitch, c->p.num_buckets);
        hipLaunchKernelGGL(bmf::bmf_zeros_kernel, dim3((unsigned)std::min<uint64_t>(n_rows, 1u << 22)), dim3(bmf::kWave), 0,
                           c->stream, c->d_rows, n_rows, pitch, c->p.num_buckets, c->d_zeros);
    }
    hipLaunchKernelGGL(bmf::bmf_qgram_ok_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, c->stream,
                       c->d_k2i, n_kmers, c->d_zeros, c->p.threshold, c->d_qgram_ok, n_words);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->loaded = true;
    return select_pruned_variant(c);
}

int bmf_load_index(bmf_ctx *c, const uint8_t *rows, uint64_t n_rows, const int32_t *kmer_to_index,
                   uint64_t n_kmers) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_load_index: null context");
    if (n_rows && !rows && !c->loaded) return fail(BMF_ERR_ARG, "rows is null");
    int rc = upload_begin(c, n_rows, kmer_to_index, n_kmers);
    if (rc != BMF_OK) return rc;
    if (n_rows) {
        const uint32_t row_bytes = (c->p.num_buckets + 7u) >> 3;
        const uint64_t piece_rows = piece_rows_of(row_bytes);
        uint8_t *d_stage = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_stage), (size_t)piece_rows * row_bytes);
        for (uint64_t first = 0; e == hipSuccess && first < n_rows; first += piece_rows) {
            e = upload_rows(c, first, std::min(piece_rows, n_rows - first), rows + (size_t)first * row_bytes, d_stage);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // one staging buffer: pageable source
        }
        (void)hipFree(d_stage);
        if (e != hipSuccess) {
            free_index(c);
            return fail(BMF_ERR_HIP, "uploading the index rows failed: %s", hipGetErrorString(e));
        }
    }
    return upload_finish(c);
}

// GPU form of bucket_indexer::index (bucket_indexer.h:49-61,170-216): see bmi_kernels.hip.h.
int bmf_build_index(bmf_ctx *c, const uint8_t *genome, uint64_t n_bases, const uint64_t *bucket_start,
                    const uint32_t *bucket_len, uint32_t n_buckets, const int32_t *kmer_to_index, uint64_t n_kmers) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_build_index: null context");
    if (c->loaded) return fail(BMF_ERR_STATE, "the q-gram index is not empty; call bmf_reset first");
    if ((n_bases && !genome) || (n_buckets && (!bucket_start || !bucket_len)) || !kmer_to_index)
        return fail(BMF_ERR_ARG, "bmf_build_index: null argument");
    const uint32_t q = c->p.q;
    if (q < 3 || q > 10) return fail(BMF_ERR_UNSUPPORTED, "the GPU index build supports 3 <= q <= 10 (got %u)", q);
    if (n_kmers != (1ull << (2 * q))) return fail(BMF_ERR_ARG, "kmer_to_index must have 4^q entries");
    if (n_buckets > c->p.num_buckets)
        return fail(BMF_ERR_ARG, "%u buckets do not fit NB = %u", n_buckets, c->p.num_buckets);
    uint32_t max_len = 0;
    for (uint32_t b = 0; b < n_buckets; b++) {
        if (bucket_start[b] > n_bases || bucket_len[b] > n_bases - bucket_start[b])
            return fail(BMF_ERR_ARG, "bucket %u lies outside the genome buffer", b);
        if (bucket_len[b] > max_len) max_len = bucket_len[b];
    }
    // rows are numbered 0..n_rows-1 in ascending q-gram hash (bucket_indexer.h:147-157)
    int64_t n_rows = 0;
    for (uint64_t i = 0; i < n_kmers; i++)
        if (kmer_to_index[i] >= 0) {
            if (kmer_to_index[i] != n_rows) return fail(BMF_ERR_ARG, "kmer_to_index must number the kept q-grams 0,1,2,... in order");
            n_rows++;
        }
    HIP_TRY(hipSetDevice(c->p.device));
    const uint32_t pitch = c->dp.pitch;
    const uint64_t n_words = ((1ull << (2 * q)) + 31) / 32;
    // Launch geometries.  A HIP grid dimension times its block dimension is a 32-bit thread count (larger grids
    // wrap silently and run a fraction of their workgroups), so the presence kernel goes out in slices of 2 Mi
    // buckets and the transpose walks the bucket groups with a strided y dimension.
    const uint32_t n_groups = (n_buckets + 63u) / 64u;
    const uint32_t tr_blocks_x = (uint32_t)((n_words / 2 + bmi::kThreads / 64 - 1) / (bmi::kThreads / 64));
    const uint32_t tr_blocks_y = std::max(1u, std::min(n_groups, 32768u));
    constexpr uint32_t kPresenceSlice = 1u << 21;
    uint8_t *d_genome = nullptr, *d_lut = c->d_lut;
    uint64_t *d_bstart = nullptr;
    uint32_t *d_blen = nullptr, *d_presence = nullptr;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    ok(dev_alloc(&c->d_rows, (size_t)(n_rows + 1) * pitch));
    ok(dev_alloc(&c->d_k2i, (size_t)n_kmers));
    ok(dev_alloc(&c->d_zeros, (size_t)n_rows));
    ok(dev_alloc(&c->d_qgram_ok, (size_t)n_words));
    ok(dev_alloc(&d_genome, (size_t)n_bases + 64));   // + slack: the presence kernel loads whole aligned 16-byte chunks
    ok(dev_alloc(&d_bstart, n_buckets));
    ok(dev_alloc(&d_blen, n_buckets));
    ok(dev_alloc(&d_presence, (size_t)n_buckets * n_words));
    if (e == hipSuccess) {
        ok(hipMemsetAsync(c->d_rows, 0, (size_t)n_rows * pitch, c->stream));
        ok(hipMemsetAsync(c->d_rows + (size_t)n_rows * pitch, 0xFF, pitch, c->stream));
        ok(hipMemcpy(d_genome, genome, (size_t)n_bases, hipMemcpyHostToDevice));
        ok(hipMemcpy(d_bstart, bucket_start, (size_t)n_buckets * sizeof(uint64_t), hipMemcpyHostToDevice));
        ok(hipMemcpy(d_blen, bucket_len, (size_t)n_buckets * sizeof(uint32_t), hipMemcpyHostToDevice));
        ok(hipMemcpy(c->d_k2i, kmer_to_index, (size_t)n_kmers * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // LDS of the presence kernel: the bitmap and (a segment of) the bucket as a 2-bit stream (+ alignment shift, + the
    // word a shift reads ahead); buckets too long for what the bitmap leaves go through in segments
    const size_t lds_room = 159 * 1024 - (size_t)n_words * 4;
    const uint32_t seg_cap = (uint32_t)std::min<size_t>(lds_room / 4 > 4 ? (lds_room / 4 - 4) * 16 : 0, 1u << 20);
    const uint32_t seg_bases = std::max(std::min(max_len, seg_cap), q);
    const uint32_t stream_words = (seg_bases + 15u + 15u) / 16u + 2u;
    const size_t lds = (size_t)n_words * 4 + (size_t)stream_words * 4;
    if (e == hipSuccess && lds > 48 * 1024)
        ok(bmhip::raise_dynamic_lds(reinterpret_cast<const void *>(bmi::bmi_presence_kernel), lds));
    if (e == hipSuccess && n_buckets) {
        for (uint32_t b0 = 0; b0 < n_buckets; b0 += kPresenceSlice)
            hipLaunchKernelGGL(bmi::bmi_presence_kernel, dim3(std::min(kPresenceSlice, n_buckets - b0)), dim3(bmi::kThreads), lds,
                               c->stream, d_genome, d_bstart + b0, d_blen + b0, d_lut, q, stream_words, seg_bases, d_presence + (size_t)b0 * n_words);
        hipLaunchKernelGGL(bmi::bmi_transpose_kernel, dim3(tr_blocks_x, tr_blocks_y), dim3(bmi::kThreads), 0, c->stream, reinterpret_cast<const uint64_t *>(d_presence), n_buckets, q,
                           c->d_k2i, c->d_rows, pitch);
        ok(hipGetLastError());
    }
    c->n_rows = (uint64_t)n_rows;
    c->dp.ones_row = (uint32_t)n_rows;
    c->dp.n_kmers = (uint32_t)n_kmers;
    if (e == hipSuccess) {
        if (n_rows)
            hipLaunchKernelGGL(bmf::bmf_zeros_kernel, dim3((unsigned)std::min<int64_t>(n_rows, 1 << 22)), dim3(bmf::kWave), 0,
                               c->stream, c->d_rows, (uint64_t)n_rows, pitch, c->p.num_buckets, c->d_zeros);
        hipLaunchKernelGGL(bmf::bmf_qgram_ok_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_k2i, n_kmers, c->d_zeros, c->p.threshold, c->d_qgram_ok, n_words);
        ok(hipGetLastError());
        ok(hipStreamSynchronize(c->stream));
    }
    (void)hipFree(d_genome);
    (void)hipFree(d_bstart);
    (void)hipFree(d_blen);
    (void)hipFree(d_presence);
    if (e != hipSuccess) {
        free_index(c);
        return fail(BMF_ERR_HIP, "bmf_build_index: %s", hipGetErrorString(e));
    }
    c->loaded = true;
    return select_pruned_variant(c);
}

int bmf_index_download(bmf_ctx *c, uint8_t *rows_out, uint64_t *n_rows_out) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_index_download: null context");
    if (!c->loaded) return fail(BMF_ERR_STATE, "no index loaded");
    if (n_rows_out) *n_rows_out = c->n_rows;
    if (!rows_out) return BMF_OK;
    HIP_TRY(hipSetDevice(c->p.device));
    const uint32_t row_bytes = (c->p.num_buckets + 7u) >> 3;
    // packed on the device piece by piece, then flat copies (a strided 2-D copy takes microseconds per row)
    const uint64_t piece_rows = piece_rows_of(row_bytes);
    uint8_t *d_stage = nullptr;
    hipError_t e = c->n_rows ? hipMalloc(reinterpret_cast<void **>(&d_stage), (size_t)piece_rows * row_bytes) : hipSuccess;
    for (uint64_t first = 0; e == hipSuccess && first < c->n_rows; first += piece_rows) {
        const uint64_t n = std::min<uint64_t>(piece_rows, c->n_rows - first), bytes = n * row_bytes;
        hipLaunchKernelGGL(bmf::bmf_pack_rows_kernel, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_rows + (size_t)first * c->dp.pitch, d_stage, n, row_bytes, c->dp.pitch);
        e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(rows_out + (size_t)first * row_bytes, d_stage, (size_t)bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d_stage);
    if (e != hipSuccess) return fail(BMF_ERR_HIP, "bmf_index_download: %s", hipGetErrorString(e));
    return BMF_OK;
}

int bmf_load_index_files(bmf_ctx *c, const char *index_dir, const char *indicator) {
    if (!c || !index_dir || !indicator) return fail(BMF_ERR_ARG, "bmf_load_index_files: null argument");
    if (c->loaded) return fail(BMF_ERR_STATE, "the q-gram index is not empty; call bmf_reset first");
    const std::string base = std::string(index_dir) + "/" + indicator;
    const uint64_t n_kmers = 1ull << (2 * c->p.q);
    // q_gram_mapper.h:331-342: text file, 4^q integers
    std::vector<int32_t> k2i(n_kmers);
    uint64_t sampled = 0;
    {
        FILE *f = fopen((base + ".kmers_index").c_str(), "r");
        if (!f) return fail(BMF_ERR_IO, "cannot open %s.kmers_index", base.c_str());
        for (uint64_t i = 0; i < n_kmers; i++) {
            int v;
            if (fscanf(f, "%d", &v) != 1) {
                fclose(f);
                return fail(BMF_ERR_IO, "%s.kmers_index ends after %llu of %llu entries", base.c_str(),
                            (unsigned long long)i, (unsigned long long)n_kmers);
            }
            k2i[i] = v;
            if (v >= 0) sampled++;
        }
        fclose(f);
    }
    // q_gram_mapper.h:345-358: `sampled` rows of ceil(NB/8) bytes.  The file is read in pieces into two
    // page-locked buffers; the copy of one piece to HBM overlaps the read of the next.
    const uint32_t row_bytes = (c->p.num_buckets + 7u) >> 3;
    FILE *f = fopen((base + ".qgram").c_str(), "rb");
    if (!f) return fail(BMF_ERR_IO, "cannot open %s.qgram", base.c_str());
    int rc = upload_begin(c, sampled, k2i.data(), n_kmers);
    if (rc != BMF_OK) {
        fclose(f);
        return rc;
    }
    const uint64_t piece_rows = piece_rows_of(row_bytes);
    uint8_t *stage[2] = {nullptr, nullptr}, *d_stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc(reinterpret_cast<void **>(&stage[i]), (size_t)piece_rows * row_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_stage[i]), (size_t)piece_rows * row_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    uint64_t got_rows = 0;
    bool short_file = false;
    for (uint64_t first = 0, piece = 0; e == hipSuccess && first < sampled; first += piece_rows, piece++) {
        const int slot = (int)(piece & 1);
        const uint64_t n = std::min(piece_rows, sampled - first);
        if (piece >= 2) e = hipEventSynchronize(done[slot]);       // the buffer's previous copy has left it
        if (e != hipSuccess) break;
        const size_t got = fread(stage[slot], row_bytes, (size_t)n, f);
        got_rows += got;
        if (got != n) {
            short_file = true;
            break;
        }
        e = upload_rows(c, first, n, stage[slot], d_stage[slot]);
        if (e == hipSuccess) e = hipEventRecord(done[slot], c->stream);
    }
    fclose(f);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    for (int i = 0; i < 2; i++) {
        if (stage[i]) (void)hipHostFree(stage[i]);
        if (d_stage[i]) (void)hipFree(d_stage[i]);
        if (done[i]) (void)hipEventDestroy(done[i]);
    }
    if (short_file || e != hipSuccess) {
        free_index(c);
        if (short_file)
            return fail(BMF_ERR_IO, "%s.qgram holds %llu rows of %u bytes, expected %llu", base.c_str(),
                        (unsigned long long)got_rows, row_bytes, (unsigned long long)sampled);
        return fail(BMF_ERR_HIP, "uploading %s.qgram failed: %s", base.c_str(), hipGetErrorString(e));
    }
    return upload_finish(c);
}

int bmf_reset(bmf_ctx *c) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_reset: null context");
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_index(c);
    return BMF_OK;
}

int bmf_index_zeros(bmf_ctx *c, uint32_t *out_zeros) {
    if (!c || !out_zeros) return fail(BMF_ERR_ARG, "bmf_index_zeros: null argument");
    if (!c->loaded) return fail(BMF_ERR_STATE, "no index loaded");
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipMemcpy(out_zeros, c->d_zeros, (size_t)c->n_rows * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BMF_OK;
}

void bmf_batch_destroy(bmf_ctx *c, bmf_batch *b) {
    if (!b) return;
    if (c) {
        (void)hipSetDevice(c->p.device);
        (void)hipStreamSynchronize(c->stream);
    }
    release_batch(b);
    delete b;
}

static int check_windows(const bmf_ctx *c, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                         const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows) {
    if (n_windows && (!win_start || !win_len)) return fail(BMF_ERR_ARG, "win_start/win_len is null");
    // one wave per (window, orientation): 2 * n * 64 threads must stay below 2^32 (HIP's grid limit)
    if (n_windows >= (1u << 25)) return fail(BMF_ERR_ARG, "too many windows in one batch (%u, limit %u)", n_windows, (1u << 25) - 1u);
    if (n_bytes && (!bases || !quals)) return fail(BMF_ERR_ARG, "bases/quals is null");
    for (uint32_t w = 0; w < n_windows; w++) {
        if (win_len[w] > c->p.read_len)
            return fail(BMF_ERR_ARG, "window %u is %u bases long, more than read_len = %u", w, win_len[w], c->p.read_len);
        if (win_start[w] > n_bytes || win_len[w] > n_bytes - win_start[w])
            return fail(BMF_ERR_ARG, "window %u [%llu, +%u) lies outside the %llu-byte read buffer", w,
                        (unsigned long long)win_start[w], win_len[w], (unsigned long long)n_bytes);
    }
    return BMF_OK;
}

// (Re)sizes the device buffers of a batch of n windows over n_bytes of reads (0: the reads live elsewhere).
static hipError_t batch_reserve(bmf_ctx *c, bmf_batch *b, size_t n, size_t n_bytes) {
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    if (n_bytes) {   // + slack: the sample kernel loads the aligned 16-byte chunks that cover a window
        ok(b->bases.need(n_bytes + 64));
        ok(b->quals.need(n_bytes + 64));
    }
    ok(b->win_start.need(n));
    ok(b->win_len.need(n));
    ok(b->lists.need(2 * n * c->dp.list_len));
    ok(b->list_n.need(n));
    ok(b->rows_anded.need(n));
    ok(b->counts.need(2 * n));
    ok(b->buckets.need(2 * n * c->p.max_candidates));
    if (c->n_slices > 1) {
        ok(b->slice_min.need(2 * n * c->n_slices));
        ok(b->slice_cnt.need(2 * n * c->n_slices));
        ok(b->slice_ids.need(2 * n * c->n_slices * c->p.max_candidates));
    }
    if (c->dp.pass1_rows || c->tunable) {
        // at their largest whatever form serves the context now: the measured forms (tune_pruned) and a later change of form
        // must never re-allocate -- DevBuf frees before it grows, and hipFree waits for every stream of the device (in the
        // tools: for the genome upload that runs beside map(), 150 ms)
        ok(b->q_counters.need(4));
        ok(b->q_slow.need(2 * n));
        ok(b->q_left.need(2 * n));
        ok(b->q_live_n.need(2 * n));
        ok(b->q_live_chunks.need(2 * n * bmf::kMaxLive));
        ok(b->q_live_mask.need(2 * n * bmf::kMaxLive));
    }
    return e;
}

// Validates the windows, (re)sizes the batch's device buffers and uploads reads + window views.
static int batch_fill(bmf_ctx *c, bmf_batch *b, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                      const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows) {
    const int rc = check_windows(c, bases, quals, n_bytes, win_start, win_len, n_windows);
    if (rc != BMF_OK) return rc;
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));   // the buffers may still be in use by an earlier run
    b->n_windows = n_windows;
    b->n_bytes = n_bytes;
    const size_t n = n_windows;
    hipError_t e = batch_reserve(c, b, n, (size_t)n_bytes);
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    if (e == hipSuccess && n_bytes) {
        ok(hipMemcpy(b->bases.p, bases, (size_t)n_bytes, hipMemcpyHostToDevice));
        ok(hipMemcpy(b->quals.p, quals, (size_t)n_bytes, hipMemcpyHostToDevice));
    }
    if (e == hipSuccess && n) {
        ok(hipMemcpy(b->win_start.p, win_start, n * sizeof(uint64_t), hipMemcpyHostToDevice));
        ok(hipMemcpy(b->win_len.p, win_len, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (e != hipSuccess) return fail(BMF_ERR_HIP, "uploading the batch failed: %s", hipGetErrorString(e));
    return BMF_OK;
}

int bmf_batch_create(bmf_ctx *c, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                     const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows, bmf_batch **out) {
    if (!c || !out) return fail(BMF_ERR_ARG, "bmf_batch_create: null argument");
    *out = nullptr;
    bmf_batch *b = new bmf_batch();
    const int rc = batch_fill(c, b, bases, quals, n_bytes, win_start, win_len, n_windows);
    if (rc != BMF_OK) {
        bmf_batch_destroy(c, b);
        return rc;
    }
    *out = b;
    return BMF_OK;
}

// Everything after the sample kernel for the first n_windows windows of `b` (row-id lists in b->lists), on the
// context's stream: the vote kernel, or the exact-pruning kernels the context currently uses.
static int launch_vote_stage(bmf_ctx *c, bmf_batch *b, uint32_t n_windows) {
    if (c->dp.pass1_rows) {
        // two-pass pruning: full-width lower-bound pass, then the queued items' exact recount (bmf_vote2.hip.h)
        const size_t n_items = 2 * (size_t)n_windows;
        if (c->sort_rows && c->dp.G >= 2) {                      // each sample's rows sparsest first
            using order_fn = void (*)(bmf::DevParams, uint32_t, const uint32_t *, const uint32_t *, uint32_t *);
            static const order_fn order[9] = {nullptr, nullptr, bmf::bmf_order_rows_kernel<2>, bmf::bmf_order_rows_kernel<3>,
                                              bmf::bmf_order_rows_kernel<4>, bmf::bmf_order_rows_kernel<5>,
                                              bmf::bmf_order_rows_kernel<6>, bmf::bmf_order_rows_kernel<7>,
                                              bmf::bmf_order_rows_kernel<8>};
            const uint64_t threads = (uint64_t)n_items * c->dp.S;
            hipLaunchKernelGGL(order[c->dp.G], dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, c->dp,
                               (uint32_t)n_items, b->list_n.p, c->d_zeros, b->lists.p);
        }
        HIP_TRY(b->q_counters.need(4));
        HIP_TRY(b->q_slow.need(2 * (size_t)b->n_windows));
        HIP_TRY(b->q_left.need(2 * (size_t)b->n_windows));
        HIP_TRY(b->q_live_n.need(2 * (size_t)b->n_windows));
        HIP_TRY(b->q_live_chunks.need(2 * (size_t)b->n_windows * bmf::kMaxLive));
        HIP_TRY(b->q_live_mask.need(2 * (size_t)b->n_windows * c->dp.max_live));
        HIP_TRY(hipMemsetAsync(b->q_counters.p, 0, 4 * sizeof(uint32_t), c->stream));
        const bmf::Pass2Queue q{b->q_counters.p, b->q_slow.p, b->q_left.p, b->q_live_n.p, b->q_live_chunks.p, b->q_live_mask.p};
        const size_t per_wave = bmf::kWave / c->dp.max_live;   // items per wave of the recount kernel
        const size_t recount_lds = per_wave * (size_t)c->dp.S * c->dp.G * sizeof(uint32_t);
        // Fixed grids: the recount walks all items (most keep a few live chunks), the slow kernel strides over the
        // queue length it reads from HBM -- nothing comes back to the host in between.
        // BMF_SLICES=n (experiment, off by default) sends the batch out in n slices with the recount of slice i on a
        // second stream under pass 1 of slice i+1.  Measured on the Egu batch: 19.1 ms in one piece, 19.3 / 19.8 /
        // 22.2 ms in 2 / 4 / 8 slices -- the recount's waves take wave slots from the bandwidth-bound pass 1, which
        // loses more than the recount's time that was to be hidden (the same outcome as overlapping the sample
        // kernel with the vote kernel in round 1).
        int n_sl = 1;
        if (const char *env = getenv("BMF_SLICES")) n_sl = std::max(1, std::min(bmf_ctx::kSlices, atoi(env)));
        if (n_sl > 1 && !c->side) {
            HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            for (auto &e : c->slice_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming));
        }
        const size_t per_slice = ((n_items + n_sl - 1) / n_sl + 1) & ~(size_t)1;   // whole windows
        for (int sl = 0; sl < n_sl; sl++) {
            const size_t first = (size_t)sl * per_slice;
            if (first >= n_items) break;
            const size_t count = std::min(per_slice, n_items - first);
            bmf::DevParams dp = c->dp;
            dp.item_base = (uint32_t)first;
            if (c->pass1_fold) {
                bmf::DevParams df = c->dpf;
                df.max_live = dp.max_live;
                df.item_base = dp.item_base;
                hipLaunchKernelGGL(c->pass1_fold, dim3((unsigned)count), dim3(bmf::kWave), 0, c->stream, df, c->d_fold, b->lists.p,
                                   b->list_n.p, b->counts.p, q);
            } else {
                hipLaunchKernelGGL(c->two_pass.pass1, dim3((unsigned)count), dim3(bmf::kWave), 0, c->stream, dp, c->d_rows, b->lists.p,
                                   b->list_n.p, b->counts.p, q);
            }
            hipStream_t rs = c->stream;
            if (n_sl > 1) {
                HIP_TRY(hipEventRecord(c->slice_done[sl], c->stream));
                HIP_TRY(hipStreamWaitEvent(c->side, c->slice_done[sl], 0));
                rs = c->side;
            }
            // One resident round of waves, each walking its share of the items: the recount kernel spills a few
            // registers to scratch, and a wave that needs scratch takes ~75 us to start (measured: 32 768 one-item waves
            // took 0.87 ms for the work 4 096 waves do in 0.3 ms) -- so no more waves than the card holds at once.
            // items with ONE stored chunk: a lane each (bmf_finish_kernel); the rest, queued by it, go through the recount
            // kernel.  (Not with BMF_SLICES: the slices' kernels overlap and would share the queue.)
            const size_t finish_lds = (size_t)bmf::kWave * ((size_t)(c->dp.S * c->dp.G) | 1u) * sizeof(uint32_t);
            const bool finish = n_sl == 1 && finish_lds <= 150 * 1024 && !c->no_finish;
            if (finish) {
                if (finish_lds > 48 * 1024)
                    HIP_TRY(bmhip::raise_dynamic_lds(reinterpret_cast<const void *>(c->two_pass.finish), finish_lds));
                hipLaunchKernelGGL(c->two_pass.finish, dim3((unsigned)((count + bmf::kWave - 1) / bmf::kWave)), dim3(bmf::kWave),
                                   finish_lds, rs, dp, c->d_rows, b->lists.p, (uint32_t)count, b->counts.p, b->buckets.p, q);
            }
            const unsigned recount_blocks = (unsigned)std::min<size_t>((count + per_wave - 1) / per_wave, c->recount_waves);
            hipLaunchKernelGGL(c->two_pass.recount, dim3(recount_blocks), dim3(bmf::kWave), recount_lds, rs, dp, c->d_rows, b->lists.p,
                               (uint32_t)count, b->counts.p, b->buckets.p, q, finish ? 1u : 0u);
        }
        if (n_sl > 1) {
            HIP_TRY(hipEventRecord(c->side_done, c->side));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->side_done, 0));
        }
        const unsigned slow_blocks = (unsigned)std::min<size_t>(n_items, 2048);
        hipLaunchKernelGGL(c->two_pass.slow, dim3(slow_blocks), dim3(bmf::kWave), 0, c->stream, c->dp, c->d_rows, b->lists.p,
                           b->counts.p, b->buckets.p, q);
        if (!c->guard_pending && n_items >= 4096) {   // the guard's sample: this run's slow-path count
            if (!c->h_guard) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_guard), 4 * sizeof(uint32_t), hipHostMallocDefault));
                HIP_TRY(hipEventCreateWithFlags(&c->guard_ev, hipEventDisableTiming));
            }
            HIP_TRY(hipMemcpyAsync(c->h_guard, b->q_counters.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(c->guard_ev, c->stream));
            c->guard_items = n_items;
            c->guard_pending = true;
        }
    } else if (c->n_slices == 1) {
        hipLaunchKernelGGL(c->vote, dim3(2 * n_windows), dim3(bmf::kWave), 0, c->stream, c->dp, c->d_rows, b->lists.p,
                           b->list_n.p, b->counts.p, b->buckets.p, (uint32_t *)nullptr);
    } else {
        hipLaunchKernelGGL(c->vote, dim3(2 * n_windows, c->n_slices), dim3(bmf::kWave), 0, c->stream, c->dp, c->d_rows,
                           b->lists.p, b->list_n.p, b->slice_cnt.p, b->slice_ids.p, b->slice_min.p);
        hipLaunchKernelGGL(bmf::bmf_merge_slices_kernel, dim3((2 * n_windows + 255) / 256), dim3(256), 0, c->stream, c->dp,
                           2 * n_windows, c->n_slices, b->slice_min.p, b->slice_cnt.p, b->slice_ids.p, b->counts.p,
                           b->buckets.p);
    }
    return BMF_OK;
}

// One way of serving BMF_FLAG_EARLY_EXIT (all of them give the plain kernel's outputs).
struct PruneChoice {
    int kind = 0;                    // 0: plain vote kernel, 1: single-pass pruning kernel, 2: two passes
    uint32_t fold = 1, rows = 0;     // two passes: the first reads `rows` rows per sample of the index folded by `fold`
    uint32_t max_live = 32;          // two passes: lanes per item of the recount kernel
    bool sort = true;                // two passes: each sample's rows sparsest first
};

static int apply_choice(bmf_ctx *c, const PruneChoice &ch) {
    c->pass1_fold = nullptr;
    c->dp.pass1_rows = 0;
    c->dp.max_live = bmf::kMaxLive;
    c->vote = pick_vote(c->cpl, c->planes, ch.kind == 1);
    if (ch.kind != 2) return BMF_OK;
    c->two_pass = pick_vote2(c->cpl, c->planes, (int)ch.max_live);
    if (!c->two_pass.pass1) return BMF_OK;
    c->dp.max_live = ch.max_live;
    c->sort_rows = ch.sort;
    // the recount's thin pass reads entry pass1_rows of every sample: the first row pass 1 has not seen at full width
    c->dp.pass1_rows = ch.fold > 1 ? 1u : ch.rows;
    if (ch.fold > 1) return build_fold(c, ch.fold, ch.rows);
    return BMF_OK;
}

// MEASURED choice of the pruning form.  The model in select_pruned_variant prices the forms from one number, the
// density of the rows a read meets; on a real genome the rows differ a hundredfold in density, reads from repeats keep
// hundreds of chunks alive, and which form wins depends on the reads as much as on the index.  So the first batch of
// at least kTuneWindows windows is used as the benchmark: every candidate form runs on its first kTuneWindows windows
// (row-id lists already written by the sample kernel), timed with HIP events on the context's stream, and the fastest
// serves the context from then on.  All forms write the same outputs, so this costs a few tens of milliseconds once
// and nothing else.  BMF_AUTOTUNE=0 keeps the model's choice; forcing a form (BMF_PASS1_ROWS, BMF_FOLD, BMF_MAX_LIVE,
// BMF_ROW_ORDER) does too; BMF_TUNE_WINDOWS=n moves the threshold (tests); BMF_LOG_TUNE=1 prints the table.
static int tune_pruned(bmf_ctx *c, bmf_batch *b) {
    c->tuned = true;
    c->guard_baseline = -1.0;
    const bmf::DevParams &d = c->dp;
    const uint32_t n_win = std::min<uint32_t>(b->n_windows, c->tune_windows);
    const bool log = getenv("BMF_LOG_TUNE") != nullptr;
    std::vector<PruneChoice> cands;
    auto add = [&](int kind, uint32_t fold, uint32_t rows, uint32_t live, bool sort) {
        PruneChoice ch;
        ch.kind = kind; ch.fold = fold; ch.rows = rows; ch.max_live = live; ch.sort = sort;
        cands.push_back(ch);
    };
    // Unsorted forms first (the order kernel re-orders the lists in place; any order is valid for every kernel): the plain
    // kernel, the single-pass kernel and the MODEL's two-pass choice -- on rows of one density the model is right and
    // the sort is pure overhead.  Then, rows sparsest first, the forms that have won somewhere: one or two rows of the
    // index itself, two or three of its 2-fold copy, three or four of its 4-fold copy on large indexes; 16 or 32 lanes.
    add(0, 1, 0, 32, false);
    add(1, 1, 0, 32, false);
    if (c->dp.pass1_rows)
        add(2, c->pass1_fold ? c->fold : 1u, c->pass1_fold ? c->dpf.pass1_rows : c->dp.pass1_rows, c->dp.max_live, false);
    const uint32_t chunks2 = ((((d.nb + 1u) / 2u + 7u) >> 3) + 15u) / 16u, chunks4 = ((((d.nb + 3u) / 4u + 7u) >> 3) + 15u) / 16u;
    for (uint32_t r = 1; r < d.G && r <= 2; r++)
        for (uint32_t live : {16u, 32u}) add(2, 1, r, live, true);
    if (chunks2 <= 256u)
        for (uint32_t r = 2; r <= d.G && r <= 3; r++)
            for (uint32_t live : {16u, 32u}) add(2, 2, r, live, true);
    if (chunks4 <= 128u && d.n_chunks >= 256u)
        for (uint32_t r = 3; r <= d.G && r <= 4; r++)
            for (uint32_t live : {16u, 32u}) add(2, 4, r, live, true);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return fail(BMF_ERR_HIP, "tune_pruned: cannot create timing events");
    }
    const auto wall0 = std::chrono::steady_clock::now();
    // EVERY queue buffer at its largest, so that no timed run allocates inside its window ...
    int rc = BMF_OK;
    const size_t n_it = 2 * (size_t)b->n_windows;
    if (b->q_counters.need(4) != hipSuccess || b->q_slow.need(n_it) != hipSuccess || b->q_left.need(n_it) != hipSuccess ||
        b->q_live_n.need(n_it) != hipSuccess || b->q_live_chunks.need(n_it * bmf::kMaxLive) != hipSuccess ||
        b->q_live_mask.need(n_it * bmf::kMaxLive) != hipSuccess)
        rc = fail(BMF_ERR_HIP, "tune_pruned: out of device memory");
    // ... and every form timed on the lists it will meet in production: the order kernel re-orders the row-id lists IN
    // PLACE, so a form that runs without it must not be timed on lists an earlier candidate left sorted.  A copy of the
    // prefix's lists (as the sample kernel wrote them) is put back before such a run.
    const size_t list_bytes = 2 * (size_t)n_win * c->dp.list_len * sizeof(uint32_t);
    // (a buffer of the context's, kept: hipFree waits for the whole device -- in the tools for the genome upload that runs
    // beside map() -- and one such wait cost the measured batch 160 ms)
    uint32_t *saved = nullptr;
    if (rc == BMF_OK && (c->tune_lists.need(list_bytes / sizeof(uint32_t)) != hipSuccess ||
                         hipMemcpyAsync(c->tune_lists.p, b->lists.p, list_bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess))
        rc = fail(BMF_ERR_HIP, "tune_pruned: out of device memory");
    saved = c->tune_lists.p;
    bool lists_sorted = false;
    std::vector<float> ms(cands.size(), 1e30f);
    auto time_one = [&](size_t i) {
        rc = apply_choice(c, cands[i]);
        if (rc != BMF_OK) return;
        if (cands[i].kind == 2 && (!c->dp.pass1_rows || (cands[i].fold > 1 && !c->pass1_fold))) return;   // no such kernel
        if (lists_sorted && !(c->sort_rows && c->dp.pass1_rows)) {
            (void)hipMemcpyAsync(b->lists.p, saved, list_bytes, hipMemcpyDeviceToDevice, c->stream);
            lists_sorted = false;
        }
        c->guard_pending = true;                                   // (no guard samples from the tuning runs)
        (void)hipEventRecord(e0, c->stream);
        rc = launch_vote_stage(c, b, n_win);
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(BMF_ERR_HIP, "tuning run failed");
        c->guard_pending = false;
        if (c->sort_rows && c->dp.pass1_rows) lists_sorted = true;
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms[i] = std::min(ms[i], t);
    };
    for (size_t i = 0; i < cands.size() && rc == BMF_OK; i++) time_one(i);      // every form once ...
    std::vector<size_t> order(cands.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return ms[x] < ms[y]; });
    for (size_t k = 0; k < std::min<size_t>(3, order.size()) && rc == BMF_OK; k++)   // ... the three fastest twice more
        for (int rep = 0; rep < 2 && rc == BMF_OK; rep++) time_one(order[k]);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (saved) {
        if (lists_sorted) (void)hipMemcpyAsync(b->lists.p, saved, list_bytes, hipMemcpyDeviceToDevice, c->stream);   // as the sample kernel left them
        (void)hipStreamSynchronize(c->stream);
    }
    if (rc != BMF_OK) return rc;
    size_t best = 0;
    for (size_t i = 1; i < cands.size(); i++)
        if (ms[i] < ms[best]) best = i;
    if (log) {
        for (size_t i = 0; i < cands.size(); i++)
            fprintf(stderr, "[bmf] tune: kind %d fold %u rows %u lanes %u sorted %d: %.3f ms per %u windows\n", cands[i].kind,
                    cands[i].fold, cands[i].rows, cands[i].max_live, (int)cands[i].sort, ms[i], n_win);
        fprintf(stderr, "[bmf] tune: chose kind %d fold %u rows %u lanes %u sorted %d (%.1f ms spent measuring)\n", cands[best].kind,
                cands[best].fold, cands[best].rows, cands[best].max_live, (int)cands[best].sort,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
    }
    rc = apply_choice(c, cands[best]);
    // (no folded pass in the end: its copy of the index stays until bmf_reset / bmf_destroy -- freeing it here would wait for
    // every stream of the device)
    return rc;
}

// The filter's kernels for the n_windows windows of `b`, reads at d_bases / d_quals, on the context's stream.
static int launch_filter(bmf_ctx *c, bmf_batch *b, const uint8_t *d_bases, const uint8_t *d_quals, hipEvent_t *ev) {
    if (ev) HIP_TRY(hipEventRecord(ev[0], c->stream));
    {
        bmf::SampleGeom g = c->sample_geom;
        g.n_windows = b->n_windows;
        const unsigned wgs = std::min<unsigned>((b->n_windows + g.waves_per_wg - 1) / g.waves_per_wg, 2048u);
        auto fn = c->sample_bitmap_lds ? bmf::bmf_sample_kernel<true> : bmf::bmf_sample_kernel<false>;
        hipLaunchKernelGGL(fn, dim3(wgs), dim3(g.waves_per_wg * bmf::kWave), c->sample_lds, c->stream, c->dp, g, d_bases, d_quals,
                           b->win_start.p, b->win_len.p, c->d_qgram_ok, c->d_k2i, c->d_pos_table, b->lists.p, b->list_n.p,
                           b->rows_anded.p);
    }
    if (ev) HIP_TRY(hipEventRecord(ev[1], c->stream));
    if (c->tunable && !c->tuned && b->n_windows >= c->tune_windows) {
        const int rc = tune_pruned(c, b);
        if (rc != BMF_OK) return rc;
    }
    if (c->guard_pending && hipEventQuery(c->guard_ev) == hipSuccess) {
        c->guard_pending = false;
        // more than 2 % of the last run's items overflowed the recount kernel's lanes (the env: tests)
        const bool trip = (uint64_t)c->h_guard[1] * 50u > c->guard_items || getenv("BMF_GUARD_TRIP");
        if (c->tunable && c->tuned) {
            // under a measured choice the share of slow items is whatever the reads make it: the first sample after
            // tuning is the baseline, and a later run with more than twice that share (+ 2 %) means the reads have
            // changed -- measure again on the next large batch
            const double share = c->guard_items ? (double)c->h_guard[1] / (double)c->guard_items : 0.0;
            if (c->guard_baseline < 0.0) c->guard_baseline = share;
            else if (share > 2.0 * c->guard_baseline + 0.02) c->tuned = false;
        } else if (c->pass1_fold && trip && !getenv("BMF_FOLD")) {
            c->pass1_fold = nullptr;                     // the folded pass lets too much through on this index:
            c->dp.pass1_rows = c->unfolded_rows;         // back to the unfolded choice (0 = the single-pass pruning kernel)
            if (c->unfolded_rows && !getenv("BMF_MAX_LIVE")) {   // ... and to the recount form that was chosen WITH it
                c->dp.max_live = (uint32_t)c->unfolded_max_live;
                c->two_pass = pick_vote2(c->cpl, c->planes, c->unfolded_max_live);
            }
        }
    }
    const int rc = launch_vote_stage(c, b, b->n_windows);
    if (rc != BMF_OK) return rc;
    if (ev) HIP_TRY(hipEventRecord(ev[2], c->stream));
    HIP_TRY(hipGetLastError());
    return BMF_OK;
}

int bmf_batch_run(bmf_ctx *c, bmf_batch *b) {
    if (!c || !b) return fail(BMF_ERR_ARG, "bmf_batch_run: null argument");
    // q_gram_mapper.h:389-393: "The q-gram index is empty. Cannot accept query."
    if (!c->loaded) return fail(BMF_ERR_STATE, "the q-gram index is empty; cannot accept query");
    if (b->n_windows == 0) return BMF_OK;
    HIP_TRY(hipSetDevice(c->p.device));
    const bool prof = c->profiling && c->prof_n < c->prof_max;
    const int rc = launch_filter(c, b, b->bases.p, b->quals.p, prof ? &c->ev[(size_t)3 * c->prof_n] : nullptr);
    if (rc == BMF_OK && prof) c->prof_n++;
    return rc;
}

int bmf_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!free_bytes || !total_bytes) return fail(BMF_ERR_ARG, "bmf_device_memory: null argument");
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
    return BMF_OK;
}

int bmf_sync(bmf_ctx *c) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_sync: null context");
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BMF_OK;
}

int bmf_batch_download(bmf_ctx *c, bmf_batch *b, uint32_t *out_counts, uint32_t *out_buckets) {
    if (!c || !b || !out_counts || !out_buckets) return fail(BMF_ERR_ARG, "bmf_batch_download: null argument");
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t n = b->n_windows;
    if (n == 0) return BMF_OK;
    // The device buffer is dense (max_candidates slots per list) but holds < 1 id per list on average:
    // exclusive-scan the counts on the device, gather the defined ids into one compact array and copy
    // only that (2 x 4 B per window + the ids instead of 2 x max_candidates x 4 B per window).
    const size_t n_items = 2 * n;
    const uint32_t mc = c->p.max_candidates;
    HIP_TRY(b->offsets.need(n_items + 1));
    HIP_TRY(b->scan_tmp.need(bmscan::tmp_elems(n_items) * sizeof(uint32_t)));
    HIP_TRY(bmscan::exclusive_sum<uint32_t>(b->counts.p, b->offsets.p, n_items, reinterpret_cast<uint32_t *>(b->scan_tmp.p), c->stream));
    HIP_TRY(hipMemcpyAsync(out_counts, b->counts.p, n_items * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint64_t total = 0;
    for (size_t i = 0; i < n_items; i++) {
        if (out_counts[i] > mc) return fail(BMF_ERR_HIP, "device returned count %u > max_candidates", out_counts[i]);
        total += out_counts[i];
    }
    if (total == 0) return BMF_OK;
    HIP_TRY(b->compact.need((size_t)total));
    hipLaunchKernelGGL(bmf::bmf_compact_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, c->stream, b->counts.p,
                       b->offsets.p, b->buckets.p, mc, (uint32_t)n_items, b->compact.p);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> ids((size_t)total);
    HIP_TRY(hipMemcpyAsync(ids.data(), b->compact.p, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    size_t at = 0;
    for (size_t i = 0; i < n_items; i++) {
        memcpy(out_buckets + i * mc, ids.data() + at, out_counts[i] * sizeof(uint32_t));
        at += out_counts[i];
    }
    return BMF_OK;
}

int bmf_batch_rows_anded(bmf_ctx *c, bmf_batch *b, uint64_t *out) {
    if (!c || !b || !out) return fail(BMF_ERR_ARG, "bmf_batch_rows_anded: null argument");
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> tmp(b->n_windows);
    if (b->n_windows)
        HIP_TRY(hipMemcpy(tmp.data(), b->rows_anded.p, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t s = 0;
    for (uint32_t v : tmp) s += v;
    *out = s;
    return BMF_OK;
}

// ---- bmf_map_windows: host buffers in, host buffers out, copies hidden under the kernels -----------------
//
// The batch is cut into pieces of consecutive windows.  Per piece: the byte span of the read buffer its
// windows cover (not the whole buffer: with several devices each context is handed its own window range of a
// shared buffer) and the rebased window views go up on the h2d stream; sample + vote + exclusive scan +
// compaction run on the context's stream; counts and the compacted ids come back on the d2h stream.  The host
// issues piece i+1 before it waits for piece i, so uploads, kernels and downloads of neighbouring pieces overlap
// (they do when `bases` / `quals` are page-locked -- bmf_pinned_alloc; pageable memory still works, its
// copies just block the issuing thread).

static uint32_t piece_windows_of(uint32_t n_windows) {
    if (const char *e = getenv("BMF_PIECE_WINDOWS")) {
        const long v = strtol(e, nullptr, 10);
        if (v > 0) return (uint32_t)std::min<long>(v, 0x3FFFFFFF);
    }
    // at least four pieces where the batch allows it, pieces of 8 Ki ... 128 Ki windows (a piece costs the host about
    // half a millisecond of submissions and unpacking: with the pruning kernels a 64 Ki piece is 1.3 ms of GPU work)
    // (round 4: at most 32 Ki windows -- a larger call is more pieces, not larger ones: what a call pays for filling and
    // draining its pipeline is one piece's gather + upload + kernels, whatever the number of pieces behind it)
    return std::min<uint32_t>(32768u, std::max<uint32_t>(8192u, (n_windows + 3u) / 4u));
}

static int map_slots_init(bmf_ctx *c) {
    if (c->slot[0]) return BMF_OK;
    HIP_TRY(hipStreamCreateWithFlags(&c->h2d, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->d2h, hipStreamNonBlocking));
    for (auto *&sl : c->slot) {
        sl = new bmf_ctx::MapSlot();
        HIP_TRY(hipEventCreateWithFlags(&sl->uploaded, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl->ran, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl->landed, hipEventDisableTiming));
    }
    return BMF_OK;
}

static hipError_t pinned_need(void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap && *p) return hipSuccess;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 8 + 64;
    const hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
    if (e == hipSuccess) *cap = want;
    return e;
}

// ids copied back with the counts, per (window, orientation): lists hold < 1 id on average; the rare piece with
// more fetches the rest in a second copy
constexpr size_t kIdsPerItemCopied = 2;

// Uploads piece [first, first + n) and queues its kernels and its downloads.  `whole`: the read buffer is
// already in HBM as a whole (c->whole_*), window starts stay absolute.
// `qual_start` != nullptr: gather mode -- window w's bases are at bases[win_start[w]], its qualities at quals[qual_start[w]]
// (a FASTQ text): the piece's windows are copied back to back into the slot's page-locked buffers by a few host threads
// (while the piece before is on the device) and go up from there.
static unsigned gather_threads(uint32_t n_windows) {
    if (const char *e = getenv("BMF_GATHER_THREADS"))             // (tests, experiments: exactly that many)
        return (unsigned)std::max(1, std::min<int>(std::min(64u, std::max(1u, n_windows)), atoi(e)));
    static const unsigned hw = std::max(1u, std::min(6u, std::thread::hardware_concurrency() / 2u));
    return std::min(hw, std::max(1u, n_windows / 2048u));         // a thread is worth starting for a few thousand windows
}

static int map_piece_issue(bmf_ctx *c, bmf_ctx::MapSlot *sl, const uint8_t *bases, const uint8_t *quals,
                           const uint64_t *win_start, const uint32_t *win_len, uint32_t first, uint32_t n, bool whole,
                           const uint64_t *qual_start = nullptr) {
    sl->first = first;
    sl->n = n;
    uint64_t lo = ~0ull, hi = 0;
    if (qual_start) {
        lo = 0;
        for (uint32_t w = first; w < first + n; w++) hi += win_len[w];
    } else {
        for (uint32_t w = first; w < first + n; w++) {
            lo = std::min(lo, win_start[w]);
            hi = std::max(hi, win_start[w] + win_len[w]);
        }
    }
    if (whole) lo = 0;
    const size_t span = whole ? 0 : (size_t)(hi - lo);
    bmf_batch *b = &sl->dev;
    b->n_windows = n;
    b->n_bytes = span;
    HIP_TRY(batch_reserve(c, b, n, span));
    const size_t n_items = 2 * (size_t)n, mc = c->p.max_candidates;
    HIP_TRY(b->offsets.need(n_items + 1));
    HIP_TRY(sl->pack.need(1 + n_items * mc));
    HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_views), &sl->h_views_cap, (size_t)n * 12));
    sl->ids_copied = kIdsPerItemCopied * n_items;
    HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_out), &sl->h_out_cap, (n_items + 1 + sl->ids_copied) * sizeof(uint32_t)));
    uint64_t *hs = reinterpret_cast<uint64_t *>(sl->h_views);
    uint32_t *hl = reinterpret_cast<uint32_t *>(sl->h_views + (size_t)n * 8);
    memcpy(hl, win_len + first, (size_t)n * sizeof(uint32_t));
    if (qual_start) {
        uint64_t at = 0;
        for (uint32_t w = 0; w < n; w++) {
            hs[w] = at;
            at += hl[w];
        }
        HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_bases), &sl->h_bases_cap, span + 64));
        HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_quals), &sl->h_quals_cap, span + 64));
        auto gather = [&](uint32_t w0, uint32_t w1) {
            for (uint32_t w = w0; w < w1; w++) {
                memcpy(sl->h_bases + hs[w], bases + win_start[first + w], hl[w]);
                memcpy(sl->h_quals + hs[w], quals + qual_start[first + w], hl[w]);
            }
        };
        const unsigned T = gather_threads(n);
        if (T <= 1) {
            gather(0, n);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; t++) pool.emplace_back(gather, (uint32_t)((uint64_t)n * t / T), (uint32_t)((uint64_t)n * (t + 1) / T));
            gather(0, (uint32_t)((uint64_t)n / T));
            for (auto &t : pool) t.join();
        }
        if (span) {
            HIP_TRY(hipMemcpyAsync(b->bases.p, sl->h_bases, span, hipMemcpyHostToDevice, c->h2d));
            HIP_TRY(hipMemcpyAsync(b->quals.p, sl->h_quals, span, hipMemcpyHostToDevice, c->h2d));
        }
    } else {
        for (uint32_t w = 0; w < n; w++) hs[w] = win_start[first + w] - lo;
        if (span) {
            HIP_TRY(hipMemcpyAsync(b->bases.p, bases + lo, span, hipMemcpyHostToDevice, c->h2d));
            HIP_TRY(hipMemcpyAsync(b->quals.p, quals + lo, span, hipMemcpyHostToDevice, c->h2d));
        }
    }
    HIP_TRY(hipMemcpyAsync(b->win_start.p, hs, (size_t)n * 8, hipMemcpyHostToDevice, c->h2d));
    HIP_TRY(hipMemcpyAsync(b->win_len.p, hl, (size_t)n * 4, hipMemcpyHostToDevice, c->h2d));
    HIP_TRY(hipEventRecord(sl->uploaded, c->h2d));
    HIP_TRY(hipStreamWaitEvent(c->stream, sl->uploaded, 0));
    const int rc = launch_filter(c, b, whole ? c->whole_bases.p : b->bases.p, whole ? c->whole_quals.p : b->quals.p, nullptr);
    if (rc != BMF_OK) return rc;
    // The dense result buffer (max_candidates slots per list) holds < 1 id per list on average: exclusive-scan
    // the counts, gather the defined ids behind their total, copy only counts + the head of that.
    HIP_TRY(b->scan_tmp.need(bmscan::tmp_elems(n_items) * sizeof(uint32_t)));
    HIP_TRY(bmscan::exclusive_sum<uint32_t>(b->counts.p, b->offsets.p, n_items, reinterpret_cast<uint32_t *>(b->scan_tmp.p), c->stream));
    hipLaunchKernelGGL(bmf::bmf_compact_total_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, c->stream,
                       b->counts.p, b->offsets.p, b->buckets.p, (uint32_t)mc, (uint32_t)n_items, sl->pack.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl->ran, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->d2h, sl->ran, 0));
    HIP_TRY(hipMemcpyAsync(sl->h_out, b->counts.p, n_items * sizeof(uint32_t), hipMemcpyDeviceToHost, c->d2h));
    const size_t head = std::min(1 + sl->ids_copied, 1 + n_items * mc);
    HIP_TRY(hipMemcpyAsync(sl->h_out + n_items, sl->pack.p, head * sizeof(uint32_t), hipMemcpyDeviceToHost, c->d2h));
    HIP_TRY(hipEventRecord(sl->landed, c->d2h));
    return BMF_OK;
}

// Where the results of a call go: dense (max_candidates slots per list, bmf_map_windows) or packed back to back
// (bmf_map_windows_compact).
struct MapOut {
    uint32_t *counts = nullptr;
    uint32_t *dense = nullptr;     // [2n x max_candidates], or
    uint32_t *ids = nullptr;       // the lists one after the other
    uint64_t ids_cap = 0, ids_used = 0;
};

// Waits for the piece in the slot and hands its results to the caller's arrays.
static int map_piece_finish(bmf_ctx *c, bmf_ctx::MapSlot *sl, MapOut &out) {
    HIP_TRY(hipEventSynchronize(sl->landed));
    const size_t n_items = 2 * (size_t)sl->n, mc = c->p.max_candidates;
    const uint32_t *counts = sl->h_out, *ids = sl->h_out + n_items + 1;
    const size_t total = sl->h_out[n_items];
    uint64_t sum = 0;
    for (size_t i = 0; i < n_items; i++) {
        if (counts[i] > mc) return fail(BMF_ERR_HIP, "device returned count %u > max_candidates", counts[i]);
        sum += counts[i];
    }
    if (sum != total) return fail(BMF_ERR_HIP, "device returned %zu ids for counts that sum to %llu", total, (unsigned long long)sum);
    std::vector<uint32_t> rest;
    if (total > sl->ids_copied) {   // rare: more ids than came back with the counts
        rest.resize(total);
        HIP_TRY(hipMemcpy(rest.data(), sl->pack.p + 1, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
        ids = rest.data();
    }
    memcpy(out.counts + 2 * (size_t)sl->first, counts, n_items * sizeof(uint32_t));
    if (out.ids) {                  // pieces finish in order: this piece's lists follow the previous piece's
        if (out.ids_used + total > out.ids_cap)
            return fail(BMF_ERR_ARG, "ids_capacity %llu is too small (2 * n_windows * max_candidates always suffices)",
                        (unsigned long long)out.ids_cap);
        memcpy(out.ids + out.ids_used, ids, total * sizeof(uint32_t));
        out.ids_used += total;
        return BMF_OK;
    }
    uint32_t *ob = out.dense + 2 * (size_t)sl->first * mc;
    size_t at = 0;
    for (size_t i = 0; i < n_items; i++) {
        for (uint32_t t = 0; t < counts[i]; t++) ob[i * mc + t] = ids[at + t];
        at += counts[i];
    }
    return BMF_OK;
}

static int map_windows_impl(bmf_ctx *c, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                            const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows, MapOut &out,
                            const uint64_t *qual_start = nullptr);

int bmf_map_windows(bmf_ctx *c, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                    const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows,
                    uint32_t *out_counts, uint32_t *out_buckets) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_map_windows: null context");
    if (n_windows && (!out_counts || !out_buckets)) return fail(BMF_ERR_ARG, "bmf_map_windows: null output");
    MapOut out;
    out.counts = out_counts;
    out.dense = out_buckets;
    return map_windows_impl(c, bases, quals, n_bytes, win_start, win_len, n_windows, out);
}

int bmf_map_windows_compact(bmf_ctx *c, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                            const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows,
                            uint32_t *out_counts, uint32_t *out_ids, uint64_t ids_capacity, uint64_t *n_ids) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_map_windows_compact: null context");
    if (!n_ids || (n_windows && (!out_counts || (!out_ids && ids_capacity))))
        return fail(BMF_ERR_ARG, "bmf_map_windows_compact: null output");
    *n_ids = 0;
    static uint32_t none;           // ids_capacity == 0 with a null pointer: still the compact form
    MapOut out;
    out.counts = out_counts;
    out.ids = out_ids ? out_ids : &none;
    out.ids_cap = ids_capacity;
    const int rc = map_windows_impl(c, bases, quals, n_bytes, win_start, win_len, n_windows, out);
    *n_ids = out.ids_used;
    return rc;
}

int bmf_map_reserve(bmf_ctx *c, uint32_t max_windows_per_call, int text_windows) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_map_reserve: null context");
    if (max_windows_per_call == 0) return BMF_OK;
    HIP_TRY(hipSetDevice(c->p.device));
    int rc = map_slots_init(c);
    if (rc != BMF_OK) return rc;
    const size_t n = std::min(piece_windows_of(max_windows_per_call), max_windows_per_call), span = n * (size_t)c->p.read_len;
    const size_t n_items = 2 * n, mc = c->p.max_candidates;
    if (c->tunable) HIP_TRY(c->tune_lists.need(2 * (size_t)std::min<size_t>(n, c->tune_windows) * c->dp.list_len));
    for (auto *sl : c->slot) {
        bmf_batch *b = &sl->dev;
        HIP_TRY(batch_reserve(c, b, n, span));
        HIP_TRY(b->offsets.need(n_items + 1));
        HIP_TRY(sl->pack.need(1 + n_items * mc));
        HIP_TRY(b->scan_tmp.need(bmscan::tmp_elems(n_items) * sizeof(uint32_t)));
        HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_views), &sl->h_views_cap, n * 12));
        HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_out), &sl->h_out_cap, (n_items + 1 + kIdsPerItemCopied * n_items) * sizeof(uint32_t)));
        if (text_windows) {
            HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_bases), &sl->h_bases_cap, span + 64));
            HIP_TRY(pinned_need(reinterpret_cast<void **>(&sl->h_quals), &sl->h_quals_cap, span + 64));
        }
    }
    return BMF_OK;
}

int bmf_map_text_windows_compact(bmf_ctx *c, const uint8_t *text, uint64_t n_bytes, const uint64_t *seq_start,
                                 const uint64_t *qual_start, const uint32_t *win_len, uint32_t n_windows,
                                 uint32_t *out_counts, uint32_t *out_ids, uint64_t ids_capacity, uint64_t *n_ids) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_map_text_windows_compact: null context");
    if (!n_ids || (n_windows && (!out_counts || !qual_start || (!out_ids && ids_capacity))))
        return fail(BMF_ERR_ARG, "bmf_map_text_windows_compact: null argument");
    *n_ids = 0;
    static uint32_t none;
    MapOut out;
    out.counts = out_counts;
    out.ids = out_ids ? out_ids : &none;
    out.ids_cap = ids_capacity;
    const int rc = map_windows_impl(c, text, text, n_bytes, seq_start, win_len, n_windows, out, qual_start);
    *n_ids = out.ids_used;
    return rc;
}

static int map_windows_impl(bmf_ctx *c, const uint8_t *bases, const uint8_t *quals, uint64_t n_bytes,
                            const uint64_t *win_start, const uint32_t *win_len, uint32_t n_windows, MapOut &out,
                            const uint64_t *qual_start) {
    if (!c->loaded) return fail(BMF_ERR_STATE, "the q-gram index is empty; cannot accept query");
    if (n_windows == 0) return BMF_OK;
    int rc = check_windows(c, bases, quals, n_bytes, win_start, win_len, n_windows);
    if (rc == BMF_OK && qual_start) rc = check_windows(c, bases, quals, n_bytes, qual_start, win_len, n_windows);
    if (rc != BMF_OK) return rc;
    HIP_TRY(hipSetDevice(c->p.device));
    rc = map_slots_init(c);
    if (rc != BMF_OK) return rc;
    const uint32_t piece = piece_windows_of(n_windows);
    const uint32_t n_pieces = (n_windows + piece - 1) / piece;
    // Bytes the pieces' spans add up to: windows in read order cover the buffer once; windows in no particular
    // order (a permuted batch) would upload most of the buffer per piece -- then it goes up once, as a whole.
    uint64_t span_sum = 0;
    for (uint32_t p = 0; p < n_pieces; p++) {
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t w = p * piece; w < std::min(n_windows, (p + 1) * piece); w++) {
            lo = std::min(lo, win_start[w]);
            hi = std::max(hi, win_start[w] + win_len[w]);
        }
        span_sum += hi - lo;
    }
    const bool whole = !qual_start && n_pieces > 1 && span_sum > n_bytes + n_bytes / 2;
    if (whole) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(c->whole_bases.need((size_t)n_bytes + 64));
        HIP_TRY(c->whole_quals.need((size_t)n_bytes + 64));
        HIP_TRY(hipMemcpyAsync(c->whole_bases.p, bases, (size_t)n_bytes, hipMemcpyHostToDevice, c->h2d));
        HIP_TRY(hipMemcpyAsync(c->whole_quals.p, quals, (size_t)n_bytes, hipMemcpyHostToDevice, c->h2d));
    }
    // issue piece p, then finish piece p - (slots - 1): its slot is free again before piece p + 1 needs it
    constexpr uint32_t kLag = bmf_ctx::kMapSlots - 1;
    for (uint32_t p = 0; p < n_pieces + kLag && rc == BMF_OK; p++) {
        if (p < n_pieces)
            rc = map_piece_issue(c, c->slot[p % bmf_ctx::kMapSlots], bases, quals, win_start, win_len, p * piece,
                                 std::min(piece, n_windows - p * piece), whole, qual_start);
        if (rc == BMF_OK && p >= kLag) rc = map_piece_finish(c, c->slot[(p - kLag) % bmf_ctx::kMapSlots], out);
    }
    if (rc != BMF_OK) {   // leave nothing in flight that still reads the caller's buffers
        (void)hipStreamSynchronize(c->h2d);
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->d2h);
    }
    return rc;
}

int bmf_profile_begin(bmf_ctx *c, uint32_t max_runs) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_profile_begin: null context");
    HIP_TRY(hipSetDevice(c->p.device));
    while (c->ev.size() < (size_t)3 * max_runs) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    c->prof_max = max_runs;
    c->prof_n = 0;
    c->profiling = true;
    return BMF_OK;
}

int bmf_profile_end(bmf_ctx *c, uint32_t *n_runs, float *ms_sample, float *ms_vote) {
    if (!c || !n_runs) return fail(BMF_ERR_ARG, "bmf_profile_end: null argument");
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->profiling = false;
    *n_runs = c->prof_n;
    for (uint32_t i = 0; i < c->prof_n; i++) {
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, c->ev[3 * i], c->ev[3 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&b, c->ev[3 * i + 1], c->ev[3 * i + 2]));
        if (ms_sample) ms_sample[i] = a;
        if (ms_vote) ms_vote[i] = b;
    }
    return BMF_OK;
}

int bmf_pinned_alloc(size_t bytes, void **out) {
    if (!out) return fail(BMF_ERR_ARG, "bmf_pinned_alloc: null argument");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return BMF_OK;
}

void bmf_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int bmf_info(bmf_ctx *c, uint32_t *row_pitch_bytes, uint32_t *chunks_per_lane, uint32_t *planes,
             uint32_t *rows_in_flight) {
    if (!c) return fail(BMF_ERR_ARG, "bmf_info: null context");
    if (row_pitch_bytes) *row_pitch_bytes = c->dp.pitch;
    if (chunks_per_lane) *chunks_per_lane = (uint32_t)c->cpl;
    if (planes) *planes = (uint32_t)c->planes;
    if (rows_in_flight) *rows_in_flight = (uint32_t)c->depth;
    return BMF_OK;
}

int bmf_batch_pass2_counts(bmf_ctx *c, bmf_batch *b, uint32_t *recounted, uint32_t *slow) {
    if (!c || !b) return fail(BMF_ERR_ARG, "bmf_batch_pass2_counts: null argument");
    uint32_t v[2] = {0, 0};
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->dp.pass1_rows && b->q_counters.p) HIP_TRY(hipMemcpy(v, b->q_counters.p, sizeof v, hipMemcpyDeviceToHost));
    if (recounted) *recounted = v[0];
    if (slow) *slow = v[1];
    return BMF_OK;
}

int bmf_batch_live_histogram(bmf_ctx *c, bmf_batch *b, uint64_t *stored, uint64_t *lowest) {
    if (!c || !b || !stored || !lowest) return fail(BMF_ERR_ARG, "bmf_batch_live_histogram: null argument");
    for (int i = 0; i < 34; i++) stored[i] = lowest[i] = 0;
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->dp.pass1_rows || !b->q_live_n.p) return BMF_OK;
    const size_t n_items = 2 * (size_t)b->n_windows, ml = bmf::kMaxLive;
    std::vector<uint32_t> live_n(n_items);
    std::vector<uint16_t> chunks(n_items * ml);
    HIP_TRY(hipMemcpy(live_n.data(), b->q_live_n.p, n_items * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(chunks.data(), b->q_live_chunks.p, n_items * ml * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n_items; i++) {
        const uint32_t t = live_n[i];
        if (t >= bmf::kSlowBound) {                                // went to the slow kernel (from pass 1 or from the recount)
            stored[33]++;
            lowest[33]++;
            continue;
        }
        // (the finish kernel marks the single-chunk items it finished with 0: they count as "0 stored" here)
        const uint32_t n = t & 0xFFu, L = (t >> 8) & 0xFFu;
        stored[std::min<uint32_t>(n, 32)]++;
        uint32_t at_l = 0;
        for (uint32_t k = 0; k < n && k < c->dp.max_live; k++) at_l += (uint32_t)(chunks[i * c->dp.max_live + k] >> bmf::kChunkIdBits) == L;
        lowest[std::min<uint32_t>(at_l, 32)]++;
    }
    return BMF_OK;
}

int bmf_batch_recount_loads(bmf_ctx *c, bmf_batch *b, uint64_t *loads) {
    if (!c || !b || !loads) return fail(BMF_ERR_ARG, "bmf_batch_recount_loads: null argument");
    uint32_t v[4] = {0, 0, 0, 0};
    HIP_TRY(hipSetDevice(c->p.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->dp.pass1_rows && b->q_counters.p) HIP_TRY(hipMemcpy(v, b->q_counters.p, sizeof v, hipMemcpyDeviceToHost));
    *loads = v[2];
    return BMF_OK;
}

int bmf_pass1_fold(bmf_ctx *c, uint32_t *fold, uint32_t *rows) {
    if (!c || !fold || !rows) return fail(BMF_ERR_ARG, "bmf_pass1_fold: null argument");
    *fold = c->pass1_fold ? c->fold : 1u;
    *rows = c->pass1_fold ? c->dpf.pass1_rows : c->dp.pass1_rows;
    return BMF_OK;
}

int bmf_pass1_rows(bmf_ctx *c, uint32_t *out) {
    if (!c || !out) return fail(BMF_ERR_ARG, "bmf_pass1_rows: null argument");
    *out = c->dp.pass1_rows;
    return BMF_OK;
}

}  // extern "C"
