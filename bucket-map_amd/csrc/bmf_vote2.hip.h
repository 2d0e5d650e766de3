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

struct Pass2Queue {
    uint32_t *counters;         // [0] items bmf_recount_kernel has work for (statistics), [1] length of slow_items,
                                // [2] 16-byte column loads bmf_recount_kernel issued (statistics: one 64-byte sector each)
    uint32_t *slow_items;
    uint32_t *live_n;           // per item: live chunks after pass 1 (0: result already final), or kSlowItem
    uint16_t *live_chunks;      // per item: max_live chunk ids, ascending
    uint4 *live_mask;           // per live chunk: which of its 128 buckets were still below F misses after pass 1
};

// The row-id list of an item holds each sample's G rows in the order DevParams::row_order gives (the sample kernel
// writes them so): 0, G-1, then the middles.  The q-grams of a k-mer overlap -- q-gram g and g+1 share q-1 bases, so a
// bucket that holds one holds the other far more often than chance (an occurrence of the first continues into the
// second with probability 1/4) -- and the rows of NEIGHBOURING q-grams AND to much less of a filter than independent
// rows would.  The passes that read only the first few entries of a sample therefore get rows that lie far apart.

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
    // chunks that still hold a bucket with < F misses, in ascending chunk order
    uint32_t n_live = 0;
    if (FOLD == 1) {
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            uint32_t a = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) a |= alive_word<CPL, PLANES>(cnt, j, x);
            const bool live = a != 0;
            const uint64_t m = __ballot(live);
            const uint32_t at = n_live + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (live && at < P.max_live) {
                Q.live_chunks[(size_t)item * P.max_live + at] = (uint16_t)cidx[j];
                Q.live_mask[(size_t)item * P.max_live + at] =
                    make_uint4(alive_word<CPL, PLANES>(cnt, j, 0), alive_word<CPL, PLANES>(cnt, j, 1),
                               alive_word<CPL, PLANES>(cnt, j, 2), alive_word<CPL, PLANES>(cnt, j, 3));
            }
            n_live += (uint32_t)__popcll(m);
        }
    } else {
        constexpr int kSub = FOLD == 4 ? 4 : 2;         // chunks of the index under one folded chunk
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            uint32_t aw[4], mine = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) aw[x] = alive_word<CPL, PLANES>(cnt, j, x);
            bool live[kSub];
#pragma unroll
            for (int h = 0; h < kSub; h++) {
                live[h] = FOLD == 4 ? aw[h] != 0 : (aw[2 * h] | aw[2 * h + 1]) != 0;
                mine += live[h] ? 1u : 0u;
            }
            uint32_t incl = mine;                       // lanes in order, a lane's chunks in order: ascending chunk ids
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o, kWave);
                if (lane >= (uint32_t)o) incl += t;
            }
            uint32_t at = n_live + incl - mine;
#pragma unroll
            for (int h = 0; h < kSub; h++)
                if (live[h]) {
                    if (at < P.max_live) {
                        Q.live_chunks[(size_t)item * P.max_live + at] = (uint16_t)(cidx[j] * (uint32_t)kSub + (uint32_t)h);
                        Q.live_mask[(size_t)item * P.max_live + at] =
                            FOLD == 4 ? make_uint4(spread4_byte(aw[h] & 0xFFu), spread4_byte((aw[h] >> 8) & 0xFFu),
                                                   spread4_byte((aw[h] >> 16) & 0xFFu), spread4_byte(aw[h] >> 24))
                                      : make_uint4(spread2_half(aw[2 * h] & 0xFFFFu), spread2_half(aw[2 * h] >> 16),
                                                   spread2_half(aw[2 * h + 1] & 0xFFFFu), spread2_half(aw[2 * h + 1] >> 16));
                    }
                    at++;
                }
            n_live += (uint32_t)__shfl((int)incl, kWave - 1, kWave);
        }
    }
    if (lane == 0) {
        if (n_live == 0) out_counts[item] = 0;
        if (n_live > P.max_live) {
            Q.slow_items[atomicAdd(&Q.counters[1], 1u)] = item;
            n_live = kSlowItem;
        }
        Q.live_n[item] = n_live;
    }
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
// list (its item's).  The counters carry on from what they hold.
template <int PLANES>
__device__ __forceinline__ uint32_t stream_column(const DevParams &P, const uint8_t *__restrict__ rows,
                                                  const uint32_t *__restrict__ list, uint32_t g0, uint32_t r, uint32_t off,
                                                  bool act, u128 (&cnt)[PLANES], uint32_t s0, uint32_t s1) {
    const uint32_t n_rows = (s1 - s0) * r;
    u128 ring[kDepthCol];
#pragma unroll
    for (int d = 0; d < kDepthCol; d++)
#pragma unroll
        for (int x = 0; x < 4; x++) ring[d].v[x] = 0;
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
                                                           uint32_t *__restrict__ out_buckets, Pass2Queue Q) {
    extern __shared__ uint32_t lds_lists[];      // the row-id lists of the wave's items: every lane of a
    constexpr uint32_t kPerWave = kWave / LIVE;   // group reads the same entry at every step
    const uint32_t lane = threadIdx.x, grp = lane / LIVE, gl = lane % LIVE;
    const uint32_t n_ids = P.S * P.G;
    uint32_t *list = lds_lists + grp * n_ids;
    uint32_t recounted = 0, loads = 0;
    // items [item_base, item_base + n_items) of the batch
    const uint32_t item_end = P.item_base + n_items;
    for (uint32_t base = P.item_base + blockIdx.x * kPerWave; base < item_end; base += gridDim.x * kPerWave) {
        const uint32_t item = base + grp;
        uint32_t n_live = item < item_end ? Q.live_n[item] : 0u;
        if (n_live == kSlowItem) n_live = 0;
        const bool have = n_live != 0;
        if (__ballot(have) == 0) continue;       // all results are final already
        recounted += (have && gl == 0) ? 1u : 0u;
        __syncthreads();                          // (one wave per block: orders the LDS reuse between rounds)
        if (have)
            for (uint32_t i = gl; i < n_ids; i += LIVE) list[i] = row_lists[(size_t)item * P.list_len + i];
        __syncthreads();
        const bool mine = gl < n_live;
        const uint32_t chunk = mine ? Q.live_chunks[(size_t)item * LIVE + gl] : 0u;
        u128 cnt[PLANES];
        auto reset = [&](bool on) {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t bits = on ? bucket_mask(P, chunk, x) : 0u;
#pragma unroll
                for (int p = 0; p < PLANES; p++) cnt[p].v[x] = start_word<PLANES>(P, bits, p);
            }
        };
        reset(mine);
        bool act = mine;
        // First one row per sample that pass 1 has NOT seen, on its own: its miss count is an independent lower
        // bound, so a chunk that survived pass 1 by chance (probability ~1e-4 per bucket) dies here with the
        // same odds, for at most S sectors instead of G*S.  Only the buckets pass 1 left alive matter (its mask), so
        // the stream is cut in two: after F + 2 samples three chunks in four are already dead and skip the rest.
        // Not worth a dependent round of loads when there is next to nothing to kill.
        if (P.pass1_rows + 1u < P.G) {
            const bool thin = mine && n_live > 3u;
            uint4 mask = make_uint4(0, 0, 0, 0);
            if (thin) mask = Q.live_mask[(size_t)item * LIVE + gl];
            auto still_alive = [&]() {
                u128 c1[PLANES][1];
#pragma unroll
                for (int p = 0; p < PLANES; p++) c1[p][0] = cnt[p];
                return ((alive_word<1, PLANES>(c1, 0, 0) & mask.x) | (alive_word<1, PLANES>(c1, 0, 1) & mask.y) |
                        (alive_word<1, PLANES>(c1, 0, 2) & mask.z) | (alive_word<1, PLANES>(c1, 0, 3) & mask.w)) != 0;
            };
            const uint32_t s_cut = min(P.S, P.F + 2u);
            loads += stream_column<PLANES>(P, rows, list, P.pass1_rows, 1u, chunk * 16u, thin, cnt, 0u, s_cut);
            const bool more = thin && still_alive();
            loads += stream_column<PLANES>(P, rows, list, P.pass1_rows, 1u, chunk * 16u, more, cnt, s_cut, P.S);
            if (thin) act = more && still_alive();
            reset(act);
        }
        loads += stream_column<PLANES>(P, rows, list, 0u, P.G, chunk * 16u, act, cnt, 0u, P.S);
        emit_best_group<PLANES, LIVE>(P, cnt, have, item, lane, chunk, out_counts, out_buckets);
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
template <int CPL, int PLANES, int DEPTH>
__device__ __forceinline__ void vote2_item(const DevParams &P, const uint8_t *__restrict__ rows,
                                           const uint32_t *__restrict__ list, uint32_t item, uint32_t lane,
                                           uint32_t *live_chunk, uint32_t *__restrict__ out_counts,
                                           uint32_t *__restrict__ out_buckets) {
    uint32_t cidx[CPL], coff[CPL];
    bool act[CPL];
    u128 cnt[PLANES][CPL];
    full_width_slots<CPL, PLANES>(P, lane, cidx, coff, act, cnt);
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
        for (int x = 0; x < 4; x++) a |= alive_word<CPL, PLANES>(cnt, j, x);
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
        vote2_item<CPL, PLANES, DEPTH>(P, rows, row_lists + (size_t)item * P.list_len, item, threadIdx.x, live_chunk,
                                       out_counts, out_buckets);
    }
}

}  // namespace bmf
