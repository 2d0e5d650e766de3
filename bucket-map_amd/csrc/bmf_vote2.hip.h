// bmf_vote2.hip.h -- two-pass exact pruning form of the bucket vote (BMF_FLAG_EARLY_EXIT).
//
// Same outputs as bmf_vote_kernel, from a fraction of the row bytes.  It rests on one inequality: the AND
// of FEWER rows has MORE bits set, so the miss count of a bucket computed from only r of the G q-gram rows
// of every sample is a LOWER bound of its true miss count.
//
//   pass 1 : stream r rows per sample at full width (S*r rows instead of S*G).  A bucket whose lower bound
//            already reaches F is dead for good (q_gram_mapper.h:75-102: it is in no level of the filter).
//            For a 22 %-dense index and the default S=15, F=6, r=1 leaves a handful of live buckets.
//   pass 2 : the exact miss counts of the few 128-bucket chunks that still hold a live bucket, from ALL
//            S*G rows but only 16 bytes of each: lane i takes live chunk i and streams its 16-byte column
//            (up to 64 live chunks, 16-32 rows in flight), first with r+1 rows per sample, which kills most
//            chunks that survived pass 1 by chance, then with all G.  With more than 64 live chunks the
//            exact recount falls back to the predicated full-layout stream of the PRUNE kernel.
//   emit   : best_results over the exact counts (dead buckets keep a count >= F and can never be in it).
//
// r is chosen on the host from the measured density of the index rows (bmf_api.hip), so that the expected
// number of live buckets after pass 1 stays small; r == G means "no gain", and the PRUNE kernel is used.
#pragma once

#include "bmf_kernels.hip.h"

namespace bmf {

// rows in flight in the 16-byte-column recount: as many as the registers pass 1 no longer needs can hold
constexpr int depth2_for(int cpl) { return cpl >= 3 && cpl <= 5 ? 32 : 16; }

// Streams `n_rows` rows through a DEPTH-deep ring: list entry of row t is list[(t / r) * G + t % r]
// (r rows of every sample; r == G walks the whole list).  ACT: per-slot load predicate.
template <int CPL, int PLANES, int DEPTH, bool EXIT>
__device__ __forceinline__ bool stream_rows(const DevParams &P, const uint8_t *__restrict__ rows,
                                            const uint32_t *__restrict__ list, uint32_t r, const uint32_t (&coff)[CPL],
                                            const bool (&act)[CPL], u128 (&cnt)[PLANES][CPL]) {
    const uint32_t n_rows = P.S * r;
    u128 ring[DEPTH][CPL];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int j = 0; j < CPL; j++)
#pragma unroll
            for (int x = 0; x < 4; x++) ring[d][j].v[x] = 0;
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
                if (act[j]) ring[d][j] = load_chunk(rp + coff[j]);
        }
    }
    u128 bf[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++)
#pragma unroll
        for (int x = 0; x < 4; x++) bf[j].v[x] = 0xFFFFFFFFu;
    uint32_t g = 0, samples_done = 0;
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
                        if (act[j]) ring[d][j] = load_chunk(rp + coff[j]);
                }
                if (++g == r) {
                    g = 0;
                    count_misses<CPL, PLANES>(bf, cnt);
                    ++samples_done;
                    if (EXIT && samples_done >= P.F && samples_done < P.S) {
                        uint32_t alive = 0;
#pragma unroll
                        for (int j = 0; j < CPL; j++)
#pragma unroll
                            for (int x = 0; x < 4; x++) alive |= ~count_ge<CPL, PLANES>(cnt, j, x, P.F);
                        if (__ballot(alive != 0) == 0) return false;   // every bucket is dead: result is empty
                    }
                }
            }
        }
    }
    return true;
}

template <int CPL, int PLANES, int DEPTH>
__global__ __launch_bounds__(kWave) void bmf_vote2_kernel(DevParams P, const uint8_t *__restrict__ rows,
                                                         const uint32_t *__restrict__ row_lists,
                                                         const uint32_t *__restrict__ list_n,
                                                         uint32_t *__restrict__ out_counts,
                                                         uint32_t *__restrict__ out_buckets, uint32_t *__restrict__) {
    constexpr int kDepth2 = depth2_for(CPL);
    __shared__ uint32_t live_chunk[kWave];
    const uint32_t item = blockIdx.x;          // 2*window + orientation
    const uint32_t lane = threadIdx.x;
    if (list_n[item >> 1] == 0) {              // window rejected by the sample kernel
        if (lane == 0) out_counts[item] = 0;
        return;
    }
    const uint32_t *__restrict__ list = row_lists + (size_t)item * P.list_len;

    uint32_t cidx[CPL], coff[CPL];
    bool act[CPL];
    u128 cnt[PLANES][CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        cidx[j] = lane + kWave * j;
        act[j] = cidx[j] < P.n_chunks;
        coff[j] = (act[j] ? cidx[j] : P.n_chunks - 1u) * 16u;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const uint32_t dead = ~bucket_mask(P, cidx[j], x);   // non-bucket bits: saturated from the start
#pragma unroll
            for (int p = 0; p < PLANES; p++) cnt[p][j].v[x] = dead;
        }
    }

    // ---- pass 1: lower bounds of the miss counts from r rows per sample
    if (!stream_rows<CPL, PLANES, DEPTH, true>(P, rows, list, P.pass1_rows, coff, act, cnt)) {
        if (lane == 0) out_counts[item] = 0;
        return;
    }
    // chunks that still hold a bucket with < F misses, compacted in ascending chunk order
    bool live[CPL];
    uint32_t n_live = 0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        uint32_t a = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) a |= ~count_ge<CPL, PLANES>(cnt, j, x, P.F);
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
        // ---- pass 2, few live chunks: lane i recounts chunk i exactly from its 16-byte column of ALL rows
        __syncthreads();
        const bool mine = lane < n_live;
        uint32_t c1[1] = {mine ? live_chunk[lane] : 0u};
        uint32_t off1[1] = {c1[0] * 16u};
        bool act1[1] = {mine};
        u128 cnt1[PLANES][1];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const uint32_t dead = mine ? ~bucket_mask(P, c1[0], x) : 0xFFFFFFFFu;
#pragma unroll
            for (int p = 0; p < PLANES; p++) cnt1[p][0].v[x] = dead;
        }
        // one more row per sample first: most chunks that survived pass 1 by chance die here, for
        // (r+1)*S sectors instead of G*S
        // (not worth a dependent round of loads when there is next to nothing to kill)
        const uint32_t r2 = P.pass1_rows + 1u;
        if (r2 < P.G && n_live > 3u) {
            stream_rows<1, PLANES, kDepth2, false>(P, rows, list, r2, off1, act1, cnt1);
            uint32_t a = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) a |= ~count_ge<1, PLANES>(cnt1, 0, x, P.F);
            act1[0] = mine && a != 0;
            if (__ballot(act1[0]) == 0) {
                if (lane == 0) out_counts[item] = 0;
                return;
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t dead = act1[0] ? ~bucket_mask(P, c1[0], x) : 0xFFFFFFFFu;
#pragma unroll
                for (int p = 0; p < PLANES; p++) cnt1[p][0].v[x] = dead;
            }
        }
        stream_rows<1, PLANES, kDepth2, false>(P, rows, list, P.G, off1, act1, cnt1);
        emit_best<1, PLANES, false>(P, cnt1, item, lane, c1, out_counts, out_buckets, nullptr);
        return;
    }

    // ---- pass 2, many live chunks: exact recount in the full layout, loading only the live chunks.
    // Chunks that are not live keep their pass-1 counters: every bucket in them is already at >= F.
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        if (live[j]) {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t dead = ~bucket_mask(P, cidx[j], x);
#pragma unroll
                for (int p = 0; p < PLANES; p++) cnt[p][j].v[x] = dead;
            }
        }
    }
    stream_rows<CPL, PLANES, DEPTH, false>(P, rows, list, P.G, coff, live, cnt);
    emit_best<CPL, PLANES, false>(P, cnt, item, lane, cidx, out_counts, out_buckets, nullptr);
}

}  // namespace bmf
