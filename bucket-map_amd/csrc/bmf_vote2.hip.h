// bmf_vote2.hip.h -- two-pass exact pruning form of the bucket vote (BMF_FLAG_EARLY_EXIT).
//
// Same outputs as bmf_vote_kernel, from a fraction of the row bytes.  It rests on one inequality: the AND
// of FEWER rows has MORE bits set, so the miss count of a bucket computed from only r of the G q-gram rows
// of every sample is a LOWER bound of its true miss count.
//
//   bmf_pass1_kernel    one wave per (window, orientation): streams r rows per sample at full width (S*r
//                       rows instead of S*G).  A bucket whose lower bound already reaches F is dead for good
//                       (q_gram_mapper.h:75-102: it is in no level of the filter).  No live bucket: the
//                       result is empty, done.  Otherwise the ids of the 128-bucket chunks that still hold a
//                       live bucket go to HBM (ascending, <= kMaxLive of them) with their number.
//   bmf_recount_kernel  LIVE (16 or 32) lanes per item, 64 / LIVE items per wave: lane i recounts live chunk i
//                       exactly from its 16-byte column of ALL S*G rows -- after one more lower bound from a
//                       row per sample that pass 1 has not seen, which kills most chunks that survived
//                       pass 1 by chance for S sectors instead of S*G -- and the
//                       group emits best_results over the exact counts.  Small register footprint, so many
//                       waves hide the latency of these short dependent streams.
//   bmf_vote2_slow_kernel  the rare items with more than kMaxLive live chunks: pass 1 again, then the exact
//                       recount at full width with loads predicated per chunk (or one lane per chunk up to 64).
//
// Almost every item keeps a few live chunks after pass 1 (a handful of unrelated buckets survive by chance),
// so the recount kernel simply walks all items and reads their live counts -- a queue would cost one atomic
// on one address per item, which alone is slower than pass 1.  Only the rare slow items are queued; their
// kernel runs a fixed grid and strides over the queue length it reads from HBM, so nothing is read back by
// the host between the launches.
//
// r is chosen on the host from the measured density of the index rows (bmf_api.hip), so that the expected
// number of live buckets after pass 1 stays small; when no r < G pays, the single-pass PRUNE kernel is used.
#pragma once

#include "bmf_kernels.hip.h"

namespace bmf {

constexpr int kMaxLive = 32;    // most live chunks (= lanes) an item can bring to the recount kernel; the kernel
                                // comes in a 16-lane form too (4 items per wave: its instruction stream is per wave,
                                // and only the live chunks' lanes do useful work) -- DevParams::max_live picks
constexpr int kDepthCol = 16;   // rows in flight in a 16-byte-column stream

constexpr uint32_t kSlowItem = 0xFFFFFFFFu;   // live_n of an item that went to the slow queue
constexpr uint32_t kSlowBound = 0xFFFFFF00u;  // ... from the recount kernel, which found an upper bound U of the best count: | U

struct Pass2Queue {
    uint32_t *counters;         // [0] items the finish / recount kernels had work for (statistics), [1] length of slow_items,
                                // [2] 16-byte column loads they issued (statistics: one 64-byte sector each), [3] length of left_items
    uint32_t *slow_items;
    uint32_t *left_items;       // items bmf_finish_kernel left to bmf_recount_kernel (more than one stored chunk)
    uint32_t *live_n;           // per item: 0 (result already final), kSlowItem, or n | L << 8 | T << 16 -- n chunks are
                                // stored, every chunk whose LEVEL (lowest pass-1 counter among its buckets) is <= T; L is
                                // the item's lowest level (counters biased as below, so 2^PLANES - 1 is "F misses or more")
    uint16_t *live_chunks;      // per item: max_live entries, ascending chunk ids: id | level << 10
    uint4 *live_mask;           // per stored chunk: which of its 128 buckets had a pass-1 counter <= T
};

// THE LEVELS.  best_results wants the buckets with the MINIMUM miss count m* (if m* < F), so a bucket whose lower bound
// from pass 1 exceeds m* is as dead as one at F -- and m* is small whenever the read really comes from somewhere
// (0 or 1 for most reads on their true strand, a few for a read whose reverse complement meets an inverted copy of a
// repeat).  On a repetitive genome that is the difference between "every copy of the repeat family is still below F"
// and "the read's own bucket and its recent duplicates".  m* is not known after pass 1, but an upper bound is cheap:
// the exact count of the chunks at the item's lowest level L (usually one chunk, the read's own).  So the recount
// kernel works in two rounds -- (A) the level-L chunks exactly, giving U = min(their best count, F - 1) >= m*;
// (B) the chunks with L < level <= U, thin pass first, judged against U instead of F -- and pass 1, when more chunks
// are below F than the recount kernel has lanes, keeps those up to the highest level T that fits; only an item whose
// U turns out above its T (a needed chunk was not kept) or whose level-L chunks alone do not fit goes to the slow kernel.
constexpr uint32_t kChunkIdBits = 10;   // a slice has at most 512 chunks of 128 buckets

// lowest counter among the buckets of `words` 32-bit words (bits that are no bucket are saturated and lose)
template <int PLANES, int NW>
__device__ __forceinline__ uint32_t min_level(const uint32_t (&c)[PLANES][NW]) {
    uint32_t cand[NW], level = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) cand[w] = 0xFFFFFFFFu;
#pragma unroll
    for (int p = PLANES - 1; p >= 0; p--) {
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) any |= cand[w] & ~c[p][w];
        const bool some = any != 0;
#pragma unroll
        for (int w = 0; w < NW; w++) cand[w] = some ? cand[w] & ~c[p][w] : cand[w];
        level |= some ? 0u : 1u << p;
    }
    return level;
}

// ... and which buckets hold it (cand), for the four words of one chunk
template <int PLANES>
__device__ __forceinline__ uint32_t min_candidates(const u128 (&cnt)[PLANES], uint32_t (&cand)[4]) {
    uint32_t level = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) cand[x] = 0xFFFFFFFFu;
#pragma unroll
    for (int p = PLANES - 1; p >= 0; p--) {
        uint32_t any = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) any |= cand[x] & ~cnt[p].v[x];
        const bool some = any != 0;
#pragma unroll
        for (int x = 0; x < 4; x++) cand[x] = some ? cand[x] & ~cnt[p].v[x] : cand[x];
        level |= some ? 0u : 1u << p;
    }
    return level;
}

// bits of one word whose PLANES-bit counter is <= T
template <int PLANES>
__device__ __forceinline__ uint32_t count_le(const uint32_t (&c)[PLANES], uint32_t T) {
    uint32_t lt = 0, eq = 0xFFFFFFFFu;
#pragma unroll
    for (int p = PLANES - 1; p >= 0; p--) {
        if ((T >> p) & 1u) {
            lt |= eq & ~c[p];
            eq &= c[p];
        } else {
            eq &= ~c[p];
        }
    }
    return lt | eq;
}

// The row-id list of an item holds each sample's G rows in the order DevParams::row_order gives (the sample kernel
// writes them so): 0, G-1, then the middles.  The q-grams of a k-mer overlap -- q-gram g and g+1 share q-1 bases, so a
// bucket that holds one holds the other far more often than chance (an occurrence of the first continues into the
// second with probability 1/4) -- and the rows of NEIGHBOURING q-grams AND to much less of a filter than independent
// rows would.  The passes that read only the first few entries of a sample therefore get rows that lie far apart.

// ... and SPARSEST FIRST (bmf_order_rows_kernel, run between the sample kernel and pass 1 when a two-pass form serves
// the index): on a real genome the rows a read meets differ widely in density (a q-gram is met in proportion to how often
// it occurs, so reads meet the dense rows), and the row with the most zeros is the one that tells the most buckets
// apart for the same bytes.  One thread per (item, sample) re-orders the sample's G entries by zeros[row], most zeros
// first; ties keep the far-apart order.  The vote ANDs all G rows and does not care; rows_anded does not change.
template <int G>
__global__ __launch_bounds__(256) void bmf_order_rows_kernel(DevParams P, uint32_t n_items, const uint32_t *__restrict__ list_n,
                                                           const uint32_t *__restrict__ zeros, uint32_t *__restrict__ row_lists) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;   // up to 2^26 items x 64 samples
    const uint32_t item = (uint32_t)(t / P.S), s = (uint32_t)(t - (uint64_t)item * P.S);
    if (item >= n_items || list_n[item >> 1] == 0) return;          // a rejected window has no lists
    uint32_t *e = row_lists + (size_t)item * P.list_len + s * G;
    uint32_t id[G], z[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        id[g] = e[g];
        z[g] = id[g] == P.ones_row ? 0u : zeros[id[g]];             // a q-gram that is not indexed tells nothing
    }
#pragma unroll
    for (int i = 1; i < G; i++)                                     // stable insertion sort, G <= 8
#pragma unroll
        for (int j = i; j > 0; j--) {
            const bool sw = z[j] > z[j - 1];
            const uint32_t zi = sw ? z[j - 1] : z[j], zj = sw ? z[j] : z[j - 1];
            const uint32_t ii = sw ? id[j - 1] : id[j], ij = sw ? id[j] : id[j - 1];
            z[j] = zi; z[j - 1] = zj; id[j] = ii; id[j - 1] = ij;
        }
#pragma unroll
    for (int g = 0; g < G; g++) e[g] = id[g];
}

// Counters in this file are BIASED: a bucket starts at 2^PLANES-1-F instead of 0, so that "F misses or more"
// is exactly "the saturating counter is all ones" -- one AND per plane instead of a bit-sliced comparison
// (the argmin and its ties do not move; emit_best is handed F = 2^PLANES-1 to match).
template <int PLANES>
__device__ __forceinline__ uint32_t start_word(const DevParams &P, uint32_t bucket_bits, int p) {
    const uint32_t bias = (1u << PLANES) - 1u - P.F;
    return ~bucket_bits | (((bias >> p) & 1u) ? 0xFFFFFFFFu : 0u);   // bits that are no bucket: saturated
}
template <int CPL, int PLANES>
__device__ __forceinline__ uint32_t alive_word(const u128 (&cnt)[PLANES][CPL], int j, int x) {
    uint32_t sat = cnt[0][j].v[x];
#pragma unroll
    for (int p = 1; p < PLANES; p++) sat &= cnt[p][j].v[x];
    return ~sat;
}
template <int PLANES>
__device__ __forceinline__ DevParams for_emit(const DevParams &P) {
    DevParams E = P;
    E.F = (1u << PLANES) - 1u;
    return E;
}

// Streams S*r rows through a DEPTH-deep ring: r rows of every sample (list entries s*G + g, g < r;
// r == G walks the whole list).  PRED: act is a per-slot load predicate (an inactive slot ANDs zeros, which
// only pushes counters that are >= F already further up); without it every slot loads (coff is clamped to
// the row, as in bmf_vote_kernel).  EXIT: stop when every bucket has >= F misses.
template <int CPL, int PLANES, int DEPTH, bool EXIT, bool PRED = true>
__device__ __forceinline__ bool stream_rows(const DevParams &P, const uint8_t *__restrict__ rows,
                                            const uint32_t *__restrict__ list, uint32_t r, const uint32_t (&coff)[CPL],
                                            const bool (&act)[CPL], u128 (&cnt)[PLANES][CPL]) {
    const uint32_t n_rows = P.S * r;
    u128 ring[DEPTH][CPL];
    if (PRED) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int j = 0; j < CPL; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) ring[d][j].v[x] = 0;
    }
    uint32_t ps = 0, pg = 0;   // prefetch cursor: sample, q-gram
    auto next_row = [&]() -> const uint8_t * {
        const uint8_t *rp = rows + (size_t)list[ps * P.G + pg] * P.pitch;
        if (++pg == r) {
            pg = 0;
            ++ps;
        }
        return rp;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        if ((uint32_t)d < n_rows) {
            const uint8_t *rp = next_row();
#pragma unroll
            for (int j = 0; j < CPL; j++)
                if (!PRED || act[j]) ring[d][j] = load_chunk(rp + coff[j]);
        }
    }
    u128 bf[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++)
#pragma unroll
        for (int x = 0; x < 4; x++) bf[j].v[x] = 0xFFFFFFFFu;
    uint32_t g = 0, samples_done = 0;
    bool check = false;
    for (uint32_t t = 0; t < n_rows; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (t + d < n_rows) {
#pragma unroll
                for (int j = 0; j < CPL; j++)
#pragma unroll
                    for (int x = 0; x < 4; x++) bf[j].v[x] &= ring[d][j].v[x];
                if (t + d + DEPTH < n_rows) {
                    const uint8_t *rp = next_row();
#pragma unroll
                    for (int j = 0; j < CPL; j++)
                        if (!PRED || act[j]) ring[d][j] = load_chunk(rp + coff[j]);
                }
                if (++g == r) {
                    g = 0;
                    count_misses<CPL, PLANES>(bf, cnt);
                    ++samples_done;
                    check = true;
                }
            }
        }
        // once per ring round (one copy of this code instead of DEPTH): is any bucket still below F misses?
        if (EXIT && check && samples_done >= P.F && samples_done < P.S) {
            uint32_t alive = 0;
#pragma unroll
            for (int j = 0; j < CPL; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) alive |= alive_word<CPL, PLANES>(cnt, j, x);
            if (__ballot(alive != 0) == 0) return false;   // every bucket is dead: result is empty
        }
        check = false;
    }
    return true;
}

// Pass 1's stream, branch-free like bmf_vote_kernel's: every slot loads (coff is clamped to the row) and the
// row ids past the last one are the all-ones row, so the compiler can wait for exactly the oldest row in
// flight and the kernel needs fewer registers than the predicated form.  The all-dead test runs once per
// ring round (one copy of that code instead of DEPTH).
template <int CPL, int PLANES, int DEPTH>
__device__ __forceinline__ bool stream_pass1(const DevParams &P, const uint8_t *__restrict__ rows,
                                             const uint32_t *__restrict__ list, uint32_t r, const uint32_t (&coff)[CPL],
                                             u128 (&cnt)[PLANES][CPL]) {
    const uint32_t n_rows = P.S * r;
    const uint32_t n_iter = (n_rows + DEPTH - 1) / DEPTH * DEPTH;
    u128 ring[DEPTH][CPL];
    uint32_t ps = 0, pg = 0;
    auto next_row = [&]() -> const uint8_t * {
        const uint32_t id = ps < P.S ? list[ps * P.G + pg] : P.ones_row;
        if (++pg == r) {
            pg = 0;
            ++ps;
        }
        return rows + (size_t)id * P.pitch;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const uint8_t *rp = next_row();
#pragma unroll
        for (int j = 0; j < CPL; j++) ring[d][j] = load_chunk(rp + coff[j]);
    }
    u128 bf[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++)
#pragma unroll
        for (int x = 0; x < 4; x++) bf[j].v[x] = 0xFFFFFFFFu;
    uint32_t g = 0, samples_done = 0;
    for (uint32_t t = 0; t < n_iter; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int j = 0; j < CPL; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) bf[j].v[x] &= ring[d][j].v[x];
            const uint8_t *rp = next_row();
#pragma unroll
            for (int j = 0; j < CPL; j++) ring[d][j] = load_chunk(rp + coff[j]);
            if (++g == r) {
                g = 0;
                count_misses<CPL, PLANES>(bf, cnt);
                ++samples_done;
            }
        }
        if (samples_done >= P.F && samples_done < P.S) {
            uint32_t alive = 0;
#pragma unroll
            for (int j = 0; j < CPL; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) alive |= alive_word<CPL, PLANES>(cnt, j, x);
            if (__ballot(alive != 0) == 0) return false;
        }
    }
    return true;
}

// Slot layout of a full-width wave and its counters at "no sample seen yet".
template <int CPL, int PLANES>
__device__ __forceinline__ void full_width_slots(const DevParams &P, uint32_t lane, uint32_t (&cidx)[CPL],
                                                 uint32_t (&coff)[CPL], bool (&act)[CPL], u128 (&cnt)[PLANES][CPL]) {
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        cidx[j] = lane + kWave * j;
        act[j] = cidx[j] < P.n_chunks;
        coff[j] = (act[j] ? cidx[j] : P.n_chunks - 1u) * 16u;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const uint32_t bits = bucket_mask(P, cidx[j], x);
#pragma unroll
            for (int p = 0; p < PLANES; p++) cnt[p][j].v[x] = start_word<PLANES>(P, bits, p);
        }
    }
}

// FOLD = 2 or 4: `rows` is the folded index (one bit per group of FOLD buckets, bmf_fold_kernel) and P its geometry (nb
// = groups, n_chunks, pitch).  A group's AND over folded rows is set whenever any of its buckets' ANDs is, so the
// group's miss count is a lower bound for each of its buckets: 1/FOLD of the bytes per row buys more rows per sample
// -- taken far apart, see above -- and far fewer chunks survive by chance.  A folded 16-byte chunk covers FOLD chunks
// of the index: FOLD = 4, each of its 32-group words is one 128-bucket chunk; FOLD = 2, each pair of words is; the
// alive bits, spread FOLD-fold, are that chunk's live-bucket mask.
template <int CPL, int PLANES, int DEPTH, int FOLD>
__device__ __forceinline__ void pass1_item(const DevParams &P, const uint8_t *__restrict__ rows,
                                           const uint32_t *__restrict__ row_lists, const uint32_t *__restrict__ list_n,
                                           uint32_t *__restrict__ out_counts, const Pass2Queue &Q) {
    const uint32_t item = P.item_base + blockIdx.x;   // 2*window + orientation
    const uint32_t lane = threadIdx.x;
    if (list_n[item >> 1] == 0) {              // window rejected by the sample kernel
        if (lane == 0) {
            out_counts[item] = 0;
            Q.live_n[item] = 0;
        }
        return;
    }
    const uint32_t *__restrict__ list = row_lists + (size_t)item * P.list_len;
    uint32_t cidx[CPL], coff[CPL];
    bool act[CPL];
    u128 cnt[PLANES][CPL];
    full_width_slots<CPL, PLANES>(P, lane, cidx, coff, act, cnt);
    if (!stream_pass1<CPL, PLANES, DEPTH>(P, rows, list, P.pass1_rows, coff, cnt)) {
        if (lane == 0) {
            out_counts[item] = 0;
            Q.live_n[item] = 0;
        }
        return;
    }
    // Level of every chunk of the index under this lane's slots: the lowest counter among its buckets (or groups).
    constexpr int kSub = FOLD == 1 ? 1 : (FOLD == 4 ? 4 : 2);      // chunks of the index under one (folded) chunk
    constexpr int kWords = 4 / kSub;                                // words of the folded chunk per chunk of the index
    constexpr uint32_t kSat = (1u << PLANES) - 1u;
    uint32_t lvl[CPL][kSub], lo = kSat;
#pragma unroll
    for (int j = 0; j < CPL; j++)
#pragma unroll
        for (int h = 0; h < kSub; h++) {
            uint32_t c[PLANES][kWords];
#pragma unroll
            for (int p = 0; p < PLANES; p++)
#pragma unroll
                for (int w = 0; w < kWords; w++) c[p][w] = cnt[p][j].v[h * kWords + w];
            lvl[j][h] = min_level<PLANES, kWords>(c);
            lo = min(lo, lvl[j][h]);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, kWave));
    if (lo == kSat) {                          // nothing below F: the result is empty
        if (lane == 0) {
            out_counts[item] = 0;
            Q.live_n[item] = 0;
        }
        return;
    }
    // How many chunks lie at or below L, L + 1, L + 2 and F - 1: the highest of these that fits the recount kernel's
    // lanes is T.  (Wave-uniform; L alone not fitting sends the item to the slow kernel.)
    const uint32_t cand_t[4] = {lo, min(lo + 1u, kSat - 1u), min(lo + 2u, kSat - 1u), kSat - 1u};
    uint32_t T = kSat, n_store = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t n = 0;
#pragma unroll
        for (int j = 0; j < CPL; j++)
#pragma unroll
            for (int h = 0; h < kSub; h++) n += (uint32_t)__popcll(__ballot(lvl[j][h] <= cand_t[i]));
        if (n <= P.max_live) {
            T = cand_t[i];
            n_store = n;
        }
    }
    if (T == kSat) {
        if (lane == 0) {
            Q.slow_items[atomicAdd(&Q.counters[1], 1u)] = item;
            Q.live_n[item] = kSlowItem;
        }
        return;
    }
    // the kept chunks in ascending order: lanes in order, a lane's sub-chunks in order, slot rounds in order
    uint32_t n_before = 0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        bool keep[kSub];
        uint32_t mine = 0;
#pragma unroll
        for (int h = 0; h < kSub; h++) {
            keep[h] = lvl[j][h] <= T;
            mine += keep[h] ? 1u : 0u;
        }
        uint32_t at, total;
        if (kSub == 1) {
            const uint64_t m = __ballot(keep[0]);
            at = n_before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            total = (uint32_t)__popcll(m);
        } else {
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o, kWave);
                if (lane >= (uint32_t)o) incl += t;
            }
            at = n_before + incl - mine;
            total = (uint32_t)__shfl((int)incl, kWave - 1, kWave);
        }
#pragma unroll
        for (int h = 0; h < kSub; h++)
            if (keep[h]) {
                uint32_t le[kWords];
#pragma unroll
                for (int w = 0; w < kWords; w++) {
                    uint32_t c[PLANES];
#pragma unroll
                    for (int p = 0; p < PLANES; p++) c[p] = cnt[p][j].v[h * kWords + w];
                    le[w] = count_le<PLANES>(c, T);
                }
                uint4 mask;
                if (FOLD == 1) mask = make_uint4(le[0], le[kWords > 1 ? 1 : 0], le[kWords > 2 ? 2 : 0], le[kWords > 3 ? 3 : 0]);
                else if (FOLD == 2) mask = make_uint4(spread2_half(le[0] & 0xFFFFu), spread2_half(le[0] >> 16),
                                                      spread2_half(le[kWords > 1 ? 1 : 0] & 0xFFFFu), spread2_half(le[kWords > 1 ? 1 : 0] >> 16));
                else mask = make_uint4(spread4_byte(le[0] & 0xFFu), spread4_byte((le[0] >> 8) & 0xFFu),
                                       spread4_byte((le[0] >> 16) & 0xFFu), spread4_byte(le[0] >> 24));
                Q.live_chunks[(size_t)item * P.max_live + at] =
                    (uint16_t)((cidx[j] * (uint32_t)kSub + (uint32_t)h) | (lvl[j][h] << kChunkIdBits));
                Q.live_mask[(size_t)item * P.max_live + at] = mask;
                at++;
            }
        n_before += total;
    }
    if (lane == 0) Q.live_n[item] = n_store | (lo << 8) | (T << 16);
}

template <int CPL, int PLANES, int DEPTH, int FOLD = 1>
__global__ __launch_bounds__(kWave) void bmf_pass1_kernel(DevParams P, const uint8_t *__restrict__ rows,
                                                         const uint32_t *__restrict__ row_lists,
                                                         const uint32_t *__restrict__ list_n,
                                                         uint32_t *__restrict__ out_counts, Pass2Queue Q) {
    pass1_item<CPL, PLANES, DEPTH, FOLD>(P, rows, row_lists, list_n, out_counts, Q);
}

// best_results (q_gram_mapper.h:90-102,471-476) over the kMaxLive lanes of one item: lane i holds the exact
// counters of one 128-bucket chunk, chunks ascending with the lane.  Group-wide steps use wave ballots
// masked to the group, so the two items of a wave need not agree on anything.
template <int PLANES, int LIVE>
__device__ __forceinline__ void emit_best_group(const DevParams &P, const u128 (&cnt)[PLANES], bool have, uint32_t item,
                                                uint32_t lane, uint32_t chunk, uint32_t *__restrict__ out_counts,
                                                uint32_t *__restrict__ out_buckets) {
    const uint32_t gl = lane % LIVE;
    const uint64_t gmask = ((1ull << LIVE) - 1ull) << (lane - gl);
    uint32_t cand[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t m_min = 0;
#pragma unroll
    for (int p = PLANES - 1; p >= 0; p--) {
        uint32_t any = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) any |= cand[x] & ~cnt[p].v[x];
        const bool some = (__ballot(any != 0) & gmask) != 0;     // a candidate with a 0 in this plane
#pragma unroll
        for (int x = 0; x < 4; x++) cand[x] = some ? cand[x] & ~cnt[p].v[x] : cand[x];
        m_min |= some ? 0u : 1u << p;
    }
    const uint32_t pc = __popc(cand[0]) + __popc(cand[1]) + __popc(cand[2]) + __popc(cand[3]);
    uint32_t incl = pc;
#pragma unroll
    for (int o = 1; o < LIVE; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, LIVE);
        if (gl >= (uint32_t)o) incl += t;
    }
    const uint32_t total = __shfl(incl, LIVE - 1, LIVE);
    if (!have) return;
    if (m_min == (1u << PLANES) - 1u || total > P.max_cand) {   // biased counters: all ones = F misses or more
        if (gl == 0) out_counts[item] = 0;
        return;
    }
    uint32_t *__restrict__ out = out_buckets + (size_t)item * P.max_cand;
    uint32_t pos = incl - pc;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        uint32_t bits = cand[x];
        while (bits) {
            out[pos++] = chunk * 128u + x * 32u + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
        }
    }
    if (gl == 0) out_counts[item] = total;
}

// Rows of a 16-byte column per lane -- rows g0 .. g0+r-1 of the samples s0 .. s1-1; every lane has its own row-id
// list (its item's).  The counters carry on from what they hold; a lane that is not `act` keeps its counters as they
// are (its ring stays all ones: no misses), and a wave without an active lane skips the stream.
template <int PLANES>
__device__ __forceinline__ uint32_t stream_column(const DevParams &P, const uint8_t *__restrict__ rows,
                                                  const uint32_t *__restrict__ list, uint32_t g0, uint32_t r, uint32_t off,
                                                  bool act, u128 (&cnt)[PLANES], uint32_t s0, uint32_t s1) {
    const uint32_t n_rows = (s1 - s0) * r;
    if (__ballot(act) == 0) return 0u;
    u128 ring[kDepthCol];
#pragma unroll
    for (int d = 0; d < kDepthCol; d++)
#pragma unroll
        for (int x = 0; x < 4; x++) ring[d].v[x] = 0xFFFFFFFFu;
    uint32_t ps = s0, pg = 0;
    auto fetch = [&](u128 &dst) {
        if (act) dst = load_chunk(rows + (size_t)list[ps * P.G + g0 + pg] * P.pitch + off);
        if (++pg == r) {
            pg = 0;
            ++ps;
        }
    };
#pragma unroll
    for (int d = 0; d < kDepthCol; d++)
        if ((uint32_t)d < n_rows) fetch(ring[d]);
    u128 bf[1];
    u128 c1[PLANES][1];
#pragma unroll
    for (int x = 0; x < 4; x++) bf[0].v[x] = 0xFFFFFFFFu;
#pragma unroll
    for (int p = 0; p < PLANES; p++) c1[p][0] = cnt[p];
    uint32_t g = 0;
    for (uint32_t t = 0; t < n_rows; t += kDepthCol) {
#pragma unroll
        for (int d = 0; d < kDepthCol; d++) {
            if (t + d < n_rows) {
#pragma unroll
                for (int x = 0; x < 4; x++) bf[0].v[x] &= ring[d].v[x];
                if (t + d + kDepthCol < n_rows) fetch(ring[d]);
                if (++g == r) {
                    g = 0;
                    count_misses<1, PLANES>(bf, c1);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PLANES; p++) cnt[p] = c1[p][0];
    return act ? n_rows : 0u;
}

// One LANE per item.  Most items leave pass 1 with exactly ONE stored chunk -- the read's own -- and in the recount kernel
// its exact count is a stream that one lane in 16 or 32 takes part in.  Here 64 such items share a wave: every lane
// streams its own item's chunk through all S*G rows (its row-id list in LDS, odd stride: the lanes of a wave hit
// different banks), finds the buckets at the minimum and writes the item's result (best_results over one chunk; the
// > max_candidates rule).  Items with more stored chunks are queued, one atomic per wave, for bmf_recount_kernel.
template <int PLANES>
__global__ __launch_bounds__(kWave) void bmf_finish_kernel(DevParams P, const uint8_t *__restrict__ rows,
                                                          const uint32_t *__restrict__ row_lists, uint32_t n_items,
                                                          uint32_t *__restrict__ out_counts, uint32_t *__restrict__ out_buckets,
                                                          Pass2Queue Q) {
    extern __shared__ uint32_t lds_lists[];
    const uint32_t lane = threadIdx.x, n_ids = P.S * P.G, stride = n_ids | 1u;
    const uint32_t item0 = P.item_base + blockIdx.x * kWave, item_end = P.item_base + n_items, item = item0 + lane;
    const uint32_t tag = item < item_end ? Q.live_n[item] : 0u;
    const uint32_t n = tag >= kSlowBound ? 0u : (tag & 0xFFu);
    const bool single = n == 1u;
    const uint64_t singles = __ballot(single);
    for (uint64_t todo = singles; todo; todo &= todo - 1ull) {       // their lists, one coalesced copy each
        const uint32_t i = (uint32_t)__builtin_ctzll(todo);
        for (uint32_t e = lane; e < n_ids; e += kWave) lds_lists[i * stride + e] = row_lists[(size_t)(item0 + i) * P.list_len + e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr uint32_t kSat = (1u << PLANES) - 1u;
    const uint32_t chunk = single ? (Q.live_chunks[(size_t)item * P.max_live] & ((1u << kChunkIdBits) - 1u)) : 0u;
    u128 cnt[PLANES];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const uint32_t bits = single ? bucket_mask(P, chunk, x) : 0u;
#pragma unroll
        for (int p = 0; p < PLANES; p++) cnt[p].v[x] = start_word<PLANES>(P, bits, p);
    }
    uint32_t loads = stream_column<PLANES>(P, rows, lds_lists + lane * stride, 0u, P.G, chunk * 16u, single, cnt, 0u, P.S);
    if (single) {
        uint32_t cand[4];
        const uint32_t level = min_candidates<PLANES>(cnt, cand);
        const uint32_t total = __popc(cand[0]) + __popc(cand[1]) + __popc(cand[2]) + __popc(cand[3]);
        const uint32_t U = min(level, kSat - 1u), lv_T = (tag >> 16) & 0xFFu;
        if (U > lv_T) {
            // the one stored chunk was all that FIT, not all there was: chunks at levels up to U were left out and may
            // hold the result -- the slow kernel takes the item, as from the recount kernel
            Q.slow_items[atomicAdd(&Q.counters[1], 1u)] = item;
            Q.live_n[item] = kSlowBound | U;
        } else if (level == kSat || total > P.max_cand) {        // nothing below F misses / more than max_candidates tie
            out_counts[item] = 0;
            Q.live_n[item] = 0;
        } else {
            uint32_t *__restrict__ out = out_buckets + (size_t)item * P.max_cand;
            uint32_t pos = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                uint32_t bits = cand[x];
                while (bits) {
                    out[pos++] = chunk * 128u + x * 32u + (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1u;
                }
            }
            out_counts[item] = total;
            Q.live_n[item] = 0;                                  // final
        }
    }
    const bool left = n >= 2u;
    const uint64_t lm = __ballot(left);
    if (lm) {
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(lm)) base = atomicAdd(&Q.counters[3], (uint32_t)__popcll(lm));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(lm), kWave);
        if (left) Q.left_items[base + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = item;
    }
    loads = wave_sum(loads);
    if (lane == 0 && singles) {
        atomicAdd(&Q.counters[0], (uint32_t)__popcll(singles));
        atomicAdd(&Q.counters[2], loads);
    }
}

// (4 waves per SIMD asked for: at most 128 VGPRs, the latency of the short dependent streams needs the waves.  The 16
//  registers this spills cost less than the waves they buy: pruned Egu step 16.96 ms; 17.35 ms at 3 waves and no spill,
//  21.3 ms at 5 waves -- tools/try_libs.sh over builds with -DBMF_RECOUNT_OCC=3 / 5.)
#ifndef BMF_RECOUNT_OCC
#define BMF_RECOUNT_OCC 4
#endif
template <int PLANES, int LIVE>
__global__ __launch_bounds__(kWave, BMF_RECOUNT_OCC) void bmf_recount_kernel(DevParams P, const uint8_t *__restrict__ rows,
                                                           const uint32_t *__restrict__ row_lists, uint32_t n_items,
                                                           uint32_t *__restrict__ out_counts,
                                                           uint32_t *__restrict__ out_buckets, Pass2Queue Q, uint32_t from_queue) {
    extern __shared__ uint32_t lds_lists[];      // the row-id lists of the wave's items: every lane of a
    constexpr uint32_t kPerWave = kWave / LIVE;   // group reads the same entry at every step
    const uint32_t lane = threadIdx.x, grp = lane / LIVE, gl = lane % LIVE;
    const uint32_t n_ids = P.S * P.G;
    uint32_t *list = lds_lists + grp * n_ids;
    uint32_t recounted = 0, loads = 0;
    // items [item_base, item_base + n_items) of the batch, or -- from_queue -- the ones bmf_finish_kernel left over
    const uint32_t first = from_queue ? 0u : P.item_base, item_end = from_queue ? Q.counters[3] : P.item_base + n_items;
    for (uint32_t base = first + blockIdx.x * kPerWave; base < item_end; base += gridDim.x * kPerWave) {
        const uint32_t at = base + grp;
        const uint32_t item = from_queue ? (at < item_end ? Q.left_items[at] : 0u) : at;
        uint32_t n_live = at < item_end ? Q.live_n[item] : 0u;
        if (n_live >= kSlowBound) n_live = 0;
        const bool have = n_live != 0;
        if (__ballot(have) == 0) continue;       // all results are final already
        recounted += (have && gl == 0) ? 1u : 0u;
        __syncthreads();                          // (one wave per block: orders the LDS reuse between rounds)
        if (have)
            for (uint32_t i = gl; i < n_ids; i += LIVE) list[i] = row_lists[(size_t)item * P.list_len + i];
        __syncthreads();
        const uint32_t lv_lo = (n_live >> 8) & 0xFFu, lv_T = (n_live >> 16) & 0xFFu;
        n_live &= 0xFFu;
        const bool mine = gl < n_live;
        const uint32_t entry = mine ? Q.live_chunks[(size_t)item * LIVE + gl] : 0u;
        const uint32_t chunk = entry & ((1u << kChunkIdBits) - 1u), level = entry >> kChunkIdBits;
        constexpr uint32_t kSat = (1u << PLANES) - 1u;
        const uint64_t gmask = ((LIVE == 64 ? 0ull : (1ull << (LIVE % 64))) - 1ull) << (lane - gl);
        u128 cnt[PLANES];
        // counters at "no sample seen": `bias` misses short of saturation (bits that are no bucket: saturated)
        auto reset = [&](bool on, uint32_t bias) {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t bits = on ? bucket_mask(P, chunk, x) : 0u;
#pragma unroll
                for (int p = 0; p < PLANES; p++) cnt[p].v[x] = ~bits | (((bias >> p) & 1u) ? 0xFFFFFFFFu : 0u);
            }
        };
        const uint32_t bias_f = kSat - P.F;          // saturated = F misses or more (start_word's bias)
        // Round A: the chunks at the item's lowest level, exactly.  Their best count bounds m* from above.
        // (Tried: no round A for an item whose lowest lower bound is already 3 misses or more -- the wrong strand of a
        // read, whose level-L chunk is an arbitrary one -- sending all its chunks through round B instead.  A quarter fewer
        // column loads and a slower kernel, 2.99 -> 5.54 ms on the uniform batch: the groups of a wave move in lockstep, a
        // wave pays for every stream any of its groups runs, and this made most waves run both.  Reverted.)
        const bool in_a = mine && level == lv_lo;
        reset(in_a, bias_f);
        loads += stream_column<PLANES>(P, rows, list, 0u, P.G, chunk * 16u, in_a, cnt, 0u, P.S);
        uint32_t best;
        {
            uint32_t c[PLANES][4];
#pragma unroll
            for (int p = 0; p < PLANES; p++)
#pragma unroll
                for (int x = 0; x < 4; x++) c[p][x] = cnt[p].v[x];
            best = min_level<PLANES, 4>(c);          // lanes outside round A are saturated
        }
#pragma unroll
        for (int o = 1; o < LIVE; o <<= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, LIVE));
        const uint32_t U = min(best, kSat - 1u);      // every bucket with a lower bound above U is dead
        // a chunk the result may need was not kept by pass 1: the slow kernel takes the item
        const bool to_slow = have && U > lv_T;
        if (to_slow && gl == 0) {
            Q.slow_items[atomicAdd(&Q.counters[1], 1u)] = item;
            Q.live_n[item] = kSlowBound | U;          // the slow kernel need not look above U either
        }
        // Round B: the other chunks with a level <= U.  First one row per sample that pass 1 has NOT seen, on its own:
        // its miss count is an independent lower bound, so a chunk that survived pass 1 by chance dies here for at most
        // S sectors instead of G*S -- judged against U, on the buckets pass 1 left at or below T (its mask), in two
        // stretches: after F + 2 samples three chunks in four are already dead and skip the rest.  Not worth a
        // dependent round of loads when there is next to nothing to kill.
        const bool in_b = mine && !to_slow && level > lv_lo && level <= U;
        bool act = in_b;
        const uint32_t n_b = (uint32_t)__popcll(__ballot(in_b) & gmask);
        if (P.pass1_rows + 1u < P.G && __ballot(in_b && n_b > 3u) != 0) {
            const bool thin = in_b && n_b > 3u;
            uint4 mask = make_uint4(0, 0, 0, 0);
            if (thin) mask = Q.live_mask[(size_t)item * LIVE + gl];
            auto still_alive = [&]() {
                u128 c1[PLANES][1];
#pragma unroll
                for (int p = 0; p < PLANES; p++) c1[p][0] = cnt[p];
                return ((alive_word<1, PLANES>(c1, 0, 0) & mask.x) | (alive_word<1, PLANES>(c1, 0, 1) & mask.y) |
                        (alive_word<1, PLANES>(c1, 0, 2) & mask.z) | (alive_word<1, PLANES>(c1, 0, 3) & mask.w)) != 0;
            };
            // biased so that saturation is "more than U misses" (U is in pass 1's biased terms: U - bias_f misses)
            if (thin) reset(true, kSat - 1u - (U - bias_f));
            const uint32_t s_cut = min(P.S, P.F + 2u);
            loads += stream_column<PLANES>(P, rows, list, P.pass1_rows, 1u, chunk * 16u, thin, cnt, 0u, s_cut);
            const bool more = thin && still_alive();
            loads += stream_column<PLANES>(P, rows, list, P.pass1_rows, 1u, chunk * 16u, more, cnt, s_cut, P.S);
            if (thin) act = more && still_alive();
        }
        if (in_b) reset(act, bias_f);                 // survivors start their exact count; the killed ones are saturated
        loads += stream_column<PLANES>(P, rows, list, 0u, P.G, chunk * 16u, act, cnt, 0u, P.S);
        emit_best_group<PLANES, LIVE>(P, cnt, have && !to_slow, item, lane, chunk, out_counts, out_buckets);
    }
    // statistics only (bmf_batch_pass2_counts): one atomic per wave, not per item
    recounted = wave_sum(recounted);
    loads = wave_sum(loads);
    if (lane == 0 && recounted) {
        atomicAdd(&Q.counters[0], recounted);
        atomicAdd(&Q.counters[2], loads);
    }
}

// One item the slow way: pass 1, then the exact recount of the live chunks -- one lane per chunk up to 64,
// else at full width with loads predicated per chunk.
// `bound`: buckets whose pass-1 counter (biased) exceeds it are dead -- 2^PLANES - 2 ("below F") unless the recount kernel
// has already found a bucket with a better exact count.
template <int CPL, int PLANES, int DEPTH>
__device__ __forceinline__ void vote2_item(const DevParams &P, const uint8_t *__restrict__ rows,
                                           const uint32_t *__restrict__ list, uint32_t item, uint32_t lane, uint32_t bound,
                                           uint32_t *live_chunk, uint32_t *__restrict__ out_counts,
                                           uint32_t *__restrict__ out_buckets, bool skip_pass1 = false) {
    uint32_t cidx[CPL], coff[CPL];
    bool act[CPL];
    u128 cnt[PLANES][CPL];
    full_width_slots<CPL, PLANES>(P, lane, cidx, coff, act, cnt);
    if (skip_pass1) {
        // pass 1 itself sent the item here: more chunks at its lowest level than the recount kernel has lanes (a read in a
        // satellite or a young repeat).  Its lower bounds would say the same again -- nearly every chunk alive -- so the exact
        // count at full width comes first and only: S*G rows instead of S*r + S*G.
        stream_rows<CPL, PLANES, DEPTH, false>(P, rows, list, P.G, coff, act, cnt);
        emit_best<CPL, PLANES, false>(for_emit<PLANES>(P), cnt, item, lane, cidx, out_counts, out_buckets, nullptr);
        return;
    }
    if (!stream_rows<CPL, PLANES, DEPTH, true>(P, rows, list, P.pass1_rows, coff, act, cnt)) {
        if (lane == 0) out_counts[item] = 0;
        return;
    }
    bool live[CPL];
    uint32_t n_live = 0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        uint32_t a = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            uint32_t c[PLANES];
#pragma unroll
            for (int p = 0; p < PLANES; p++) c[p] = cnt[p][j].v[x];
            a |= count_le<PLANES>(c, bound);
        }
        live[j] = a != 0;
        const uint64_t m = __ballot(live[j]);
        const uint32_t at = n_live + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (live[j] && at < (uint32_t)kWave) live_chunk[at] = cidx[j];
        n_live += (uint32_t)__popcll(m);
    }
    if (n_live == 0) {
        if (lane == 0) out_counts[item] = 0;
        return;
    }
    if (n_live <= (uint32_t)kWave) {
        __syncthreads();
        const bool mine = lane < n_live;
        uint32_t c1[1] = {mine ? live_chunk[lane] : 0u};
        uint32_t off1[1] = {c1[0] * 16u};
        bool act1[1] = {mine};
        u128 cnt1[PLANES][1];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const uint32_t bits = mine ? bucket_mask(P, c1[0], x) : 0u;
#pragma unroll
            for (int p = 0; p < PLANES; p++) cnt1[p][0].v[x] = start_word<PLANES>(P, bits, p);
        }
        stream_rows<1, PLANES, kDepthCol, false>(P, rows, list, P.G, off1, act1, cnt1);
        emit_best<1, PLANES, false>(for_emit<PLANES>(P), cnt1, item, lane, c1, out_counts, out_buckets, nullptr);
        __syncthreads();
        return;
    }
    // chunks that are not live keep their pass-1 counters: every bucket in them is already at >= F
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        if (live[j]) {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t bits = bucket_mask(P, cidx[j], x);
#pragma unroll
                for (int p = 0; p < PLANES; p++) cnt[p][j].v[x] = start_word<PLANES>(P, bits, p);
            }
        }
    }
    stream_rows<CPL, PLANES, DEPTH, false>(P, rows, list, P.G, coff, live, cnt);
    emit_best<CPL, PLANES, false>(for_emit<PLANES>(P), cnt, item, lane, cidx, out_counts, out_buckets, nullptr);
}

template <int CPL, int PLANES, int DEPTH>
__global__ __launch_bounds__(kWave) void bmf_vote2_slow_kernel(DevParams P, const uint8_t *__restrict__ rows,
                                                              const uint32_t *__restrict__ row_lists,
                                                              uint32_t *__restrict__ out_counts,
                                                              uint32_t *__restrict__ out_buckets, Pass2Queue Q) {
    __shared__ uint32_t live_chunk[kWave];
    const uint32_t n_slow = Q.counters[1];
    for (uint32_t i = blockIdx.x; i < n_slow; i += gridDim.x) {
        const uint32_t item = Q.slow_items[i];
        const uint32_t tag = Q.live_n[item];
        const uint32_t bound = tag != kSlowItem && tag >= kSlowBound ? (tag & 0xFFu) : (1u << PLANES) - 2u;
        vote2_item<CPL, PLANES, DEPTH>(P, rows, row_lists + (size_t)item * P.list_len, item, threadIdx.x, bound, live_chunk,
                                       out_counts, out_buckets, tag == kSlowItem);
    }
}

}  // namespace bmf
