// bml_kernels.hip.h -- gfx950 kernels of the locator candidate scan (include/bml.h).
//
//   bml_scan_kernel   : one 256-thread workgroup per (bucket, chunk of <= max_pairs candidates).  Packs the
//                       bucket to 2 bits/base in LDS, hashes the k-mers the chunk's candidates ask for
//                       into an LDS open-addressing table, scans every k-mer of the bucket against the
//                       table and emits (target, offset) occurrences -- the work the reference does by
//                       building an unordered_multimap of ALL k-mers per bucket
//                       (bucket_locator.h:162-177) and calling equal_range per sample (:246).
//   (sort)            : occurrences are sorted by (candidate, sample in processing order, offset
//                       descending) -- the order libstdc++'s equal_range yields equal keys.
//   bml_replay_kernel : one thread per candidate replays _find_offset's order-dependent vote
//                       (bucket_locator.h:233-288) over its sorted occurrences; its std::map<int,unsigned>
//                       lives as a sorted array in a scratch slice as long as the candidate's occurrences.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bm_dna4.hip.h"

namespace bml {

constexpr int kThreads = 256;        // replay kernel
constexpr int kScanThreads = 1024;   // scan kernel: its loops are chains of dependent LDS reads, and the 66 KB of
                                     // LDS per workgroup allow two workgroups per CU -- 16 waves each hide the latency
                                     // that 4 waves each did not (6.96 -> 2.70 ms per 1 M candidates)
constexpr uint32_t kTableSlots = 4096;        // LDS open-addressing table (targets of one chunk)
constexpr uint32_t kLdsOcc = 2048;            // occurrences staged in LDS per workgroup
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kFilterWords = 2048;       // 64 Kbit presence filter in front of the table: a bucket has 65 825 k-mers
                                              // and a chunk asks for at most 2 040 of all 4^k, so 97 % of the scan's
                                              // probes end at ONE LDS read instead of walking a half-full table

struct Chunk {
    uint32_t bucket;
    uint32_t pair_begin;
    uint32_t pair_count;
};

struct LocParams {
    uint32_t k, p;
    int32_t allowed_mismatch, allowed_indel;
    uint32_t max_words;        // LDS words reserved for the packed bucket
};

__host__ __device__ inline size_t scan_lds_bytes(uint32_t max_words) {
    return (size_t)kLdsOcc * 8 + 16 + (size_t)kFilterWords * 4 + (size_t)kTableSlots * 8 + ((size_t)max_words + 4) * 4;
}

__device__ __forceinline__ uint32_t filter_bit(uint32_t h) { return (h * 2246822519u) >> 16; }   // 16 bits

// utils.h:291-302
__device__ __forceinline__ uint32_t hash_reverse_complement(uint32_t h, uint32_t k) {
    uint32_t rc = 0;
    for (uint32_t i = 0; i < k; i++) {
        rc = (rc << 2) | ((~h) & 3u);
        h >>= 2;
    }
    return rc;
}

__device__ __forceinline__ uint32_t slot_of(uint32_t h) { return (h * 2654435761u) >> 20; }   // 12 bits

// LDS layout (all dynamic, 16-byte aligned base): locc[kLdsOcc] u64 | gbase u64 | lds_cnt u32 (+pad) |
//                       filter[kFilterWords] u32 | tkey[kTableSlots] u32 | ttgt[kTableSlots] u32 | packed[max_words + 4] u32
__global__ __launch_bounds__(kScanThreads) void bml_scan_kernel(
    LocParams P, const uint8_t *__restrict__ genome, const uint64_t *__restrict__ bucket_start,
    const uint32_t *__restrict__ bucket_len, const uint8_t *__restrict__ dna4_lut, const Chunk *__restrict__ chunks,
    const uint32_t *__restrict__ sample_hash, const uint32_t *__restrict__ pair_window,
    const uint8_t *__restrict__ pair_rc, uint64_t *__restrict__ occ_keys, unsigned long long *__restrict__ occ_count,
    unsigned long long occ_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint64_t *locc = reinterpret_cast<uint64_t *>(smem);
    unsigned long long &gbase = *reinterpret_cast<unsigned long long *>(locc + kLdsOcc);
    uint32_t &lds_cnt = *reinterpret_cast<uint32_t *>(locc + kLdsOcc + 1);
    uint32_t *filter = reinterpret_cast<uint32_t *>(locc + kLdsOcc + 2);
    uint32_t *tkey = filter + kFilterWords;
    uint32_t *ttgt = tkey + kTableSlots;
    uint32_t *packed = ttgt + kTableSlots;
    (void)dna4_lut;

    const Chunk ch = chunks[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t nb = bucket_len[ch.bucket];
    const uint64_t start = bucket_start[ch.bucket];

    if (tid == 0) lds_cnt = 0;
    for (uint32_t s = tid; s < kTableSlots; s += kScanThreads) ttgt[s] = kEmpty;
    for (uint32_t s = tid; s < kFilterWords; s += kScanThreads) filter[s] = 0;

    // 2-bit packing, first base in the most significant bits of each word: one aligned 16-byte load per stream
    // word (the bucket may start at any byte: the stream starts at the aligned chunk that holds its first base,
    // base j of the bucket is stream position shift + j; the chunks before the genome buffer's first byte do not
    // exist, and hipMalloc'd buffers are 256-byte aligned, so the aligned start is inside the buffer)
    const uint32_t shift = (uint32_t)(start & 15u);
    const uint8_t *abase = genome + (start - shift);
    const uint32_t n_words = (shift + nb + 15u) / 16u;
    for (uint32_t w = tid; w < n_words + 2u; w += kScanThreads) {
        uint32_t word = 0;
        if (w < n_words) {
            const uint4 v = *reinterpret_cast<const uint4 *>(abase + 16u * w);
            word = bmdna::dna4_pack16(v);
        }
        packed[w] = word;
    }
    __syncthreads();
    // the k-mers this chunk's candidates ask for: target t = (candidate, i-th processed sample)
    const uint32_t n_t = ch.pair_count * P.p;
    for (uint32_t t = tid; t < n_t; t += kScanThreads) {
        const uint32_t pair = ch.pair_begin + t / P.p, i = t % P.p;
        const uint32_t w = pair_window[pair];
        const bool rc = pair_rc[pair] != 0;
        // bucket_locator.h:235-242: reverse-complement candidates start from the last sample
        uint32_t h = sample_hash[(size_t)w * P.p + (rc ? P.p - 1u - i : i)];
        if (rc) h = hash_reverse_complement(h, P.k);
        uint32_t slot = slot_of(h);
        while (atomicCAS(&ttgt[slot], kEmpty, t) != kEmpty) slot = (slot + 1u) & (kTableSlots - 1u);
        tkey[slot] = h;
        const uint32_t fb = filter_bit(h);
        atomicOr(&filter[fb >> 5], 1u << (fb & 31u));
    }
    __syncthreads();

    // scan every k-mer of the bucket (bucket_locator.h:172-176 enumerates the same k-mers)
    const uint32_t nk = nb >= P.k ? nb - P.k + 1u : 0u;
    const uint32_t kmask = P.k >= 16 ? 0xFFFFFFFFu : ((1u << (2u * P.k)) - 1u);
    for (uint32_t j = tid; j < nk; j += kScanThreads) {
        const uint32_t at = shift + j;
        const uint64_t two = ((uint64_t)packed[at >> 4] << 32) | packed[(at >> 4) + 1];
        const uint32_t h = (uint32_t)(two >> (64u - 2u * (at & 15u) - 2u * P.k)) & kmask;
        const uint32_t fb = filter_bit(h);
        if (!((filter[fb >> 5] >> (fb & 31u)) & 1u)) continue;          // nobody asked for this k-mer
        uint32_t slot = slot_of(h), t;
        while ((t = ttgt[slot]) != kEmpty) {
            if (tkey[slot] == h) {
                const uint32_t target = (ch.pair_begin + t / P.p) * P.p + t % P.p;
                const uint64_t key = ((uint64_t)target << 32) | (uint32_t)(0x7FFFFFFFu - j);   // offset descending
                const uint32_t at_l = atomicAdd(&lds_cnt, 1u);
                if (at_l < kLdsOcc) {
                    locc[at_l] = key;
                } else {   // rare (repeats): past the LDS staging area, go to HBM directly
                    const unsigned long long g = atomicAdd(occ_count, 1ull);
                    if (g < occ_cap) occ_keys[g] = key;
                }
            }
            slot = (slot + 1u) & (kTableSlots - 1u);
        }
    }
    __syncthreads();
    const uint32_t n_l = lds_cnt < kLdsOcc ? lds_cnt : kLdsOcc;
    if (tid == 0 && n_l) gbase = atomicAdd(occ_count, (unsigned long long)n_l);
    __syncthreads();
    for (uint32_t i = tid; i < n_l; i += kScanThreads)
        if (gbase + i < occ_cap) occ_keys[gbase + i] = locc[i];
}

__device__ __forceinline__ uint64_t lower_bound_key(const uint64_t *keys, uint64_t n, uint64_t v) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// _find_offset (bucket_locator.h:209-290) for one candidate per thread.
__global__ __launch_bounds__(kThreads) void bml_replay_kernel(
    LocParams P, const uint64_t *__restrict__ keys, uint64_t n_occ, const uint16_t *__restrict__ sample_pos,
    const uint32_t *__restrict__ seg_len, const uint32_t *__restrict__ pair_window, const uint8_t *__restrict__ pair_rc,
    uint32_t n_pairs, int32_t *__restrict__ prop_key, uint32_t *__restrict__ prop_votes, int32_t *__restrict__ out_offset,
    uint32_t *__restrict__ out_votes) {
    const uint32_t pair = blockIdx.x * kThreads + threadIdx.x;
    if (pair >= n_pairs) return;
    const uint64_t first = lower_bound_key(keys, n_occ, (uint64_t)pair * P.p << 32);
    const uint64_t end = lower_bound_key(keys, n_occ, (uint64_t)(pair + 1u) * P.p << 32);
    int32_t *pk = prop_key + first;      // the vote_counter map: sorted keys ...
    uint32_t *pv = prop_votes + first;   // ... and their votes; never more entries than occurrences
    uint32_t np = 0;
    const uint32_t w = pair_window[pair];
    const bool rc = pair_rc[pair] != 0;
    const uint32_t length = seg_len[w];
    uint64_t cur = first;
    for (uint32_t i = 0; i < P.p; i++) {
        uint32_t idx = sample_pos[(size_t)w * P.p + (rc ? P.p - 1u - i : i)];
        if (rc) idx = length - P.k - idx;                        // :242 where the k-mer starts on the other strand
        const uint32_t target = pair * P.p + i;
        const bool was_empty = np == 0;                          // :247 evaluated once per sample
        while (cur < end && (uint32_t)(keys[cur] >> 32) == target) {
            const uint32_t occ = 0x7FFFFFFFu - (uint32_t)keys[cur];
            const int32_t position = (int32_t)(occ - idx);       // :250,257 (unsigned arithmetic, wraps like int)
            // lower_bound(position - indel) .. upper_bound(position + indel)  (:259-260)
            uint32_t lb = 0, ub = np;
            if (!was_empty) {
                const int32_t lo_key = position - P.allowed_indel, hi_key = position + P.allowed_indel;
                uint32_t a = 0, b = np;
                while (a < b) { const uint32_t m = (a + b) / 2; if (pk[m] < lo_key) a = m + 1; else b = m; }
                lb = a;
                b = np;
                while (a < b) { const uint32_t m = (a + b) / 2; if (pk[m] <= hi_key) a = m + 1; else b = m; }
                ub = a;
            }
            if (!was_empty && lb < ub) {
                for (uint32_t v = lb; v < ub; v++) pv[v]++;     // :262-265 every proposal in range gets the vote
            } else {
                // vote_counter[position]++ (:251,269): insert, or increment when the key exists
                uint32_t a = 0, b = np;
                while (a < b) { const uint32_t m = (a + b) / 2; if (pk[m] < position) a = m + 1; else b = m; }
                if (a < np && pk[a] == position) {
                    pv[a]++;
                } else {
                    for (uint32_t v = np; v > a; v--) { pk[v] = pk[v - 1]; pv[v] = pv[v - 1]; }
                    pk[a] = position;
                    pv[a] = 1;
                    np++;
                }
            }
            cur++;
        }
    }
    int32_t off = -1;
    uint32_t votes = 0;
    if (np) {
        uint32_t best = 0;                                       // :281-283 most votes, ties -> smallest offset
        for (uint32_t v = 1; v < np; v++)
            if (pv[v] > pv[best]) best = v;
        // :284 unsigned >= int compares as unsigned
        if (pv[best] >= (uint32_t)((int32_t)P.p - P.allowed_mismatch) && pk[best] >= 0) {
            off = pk[best];
            votes = pv[best];
        }
    }
    out_offset[pair] = off;
    out_votes[pair] = votes;
}

// --------------------------------------------------------------------------------------------------
// _prepare_read_query (bucket_locator.h:292-347): the p (hash, position) pairs the locator asks of every
// window.  One wave per window.  LDS (dynamic): lut[256] | code[max_len] | qrank[max_len] | pad | good[max_nk] u16
// --------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t sample_lds_bytes(uint32_t max_len, uint32_t k) {
    const uint32_t max_nk = max_len >= k ? max_len - k + 1 : 1;
    return ((256 + 2 * (size_t)max_len + 3) & ~(size_t)3) + 2 * (size_t)max_nk;
}

__global__ __launch_bounds__(64) void bml_sample_kernel(uint32_t k, uint32_t p, uint32_t minq, uint32_t max_len,
                                                       const uint8_t *__restrict__ bases, const uint8_t *__restrict__ quals,
                                                       const uint64_t *__restrict__ win_start,
                                                       const uint32_t *__restrict__ win_len,
                                                       const uint8_t *__restrict__ dna4_lut,
                                                       const uint16_t *__restrict__ pos_table,
                                                       uint32_t *__restrict__ out_hash, uint16_t *__restrict__ out_pos,
                                                       uint8_t *__restrict__ out_has) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sample_smem[];
    uint8_t *lut = sample_smem;
    uint8_t *code = sample_smem + 256;
    uint8_t *qrank = code + max_len;
    uint16_t *good = reinterpret_cast<uint16_t *>(sample_smem + ((256 + 2 * (size_t)max_len + 3) & ~(size_t)3));
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    const uint64_t off = win_start[w];
    const uint32_t len = win_len[w];
    reinterpret_cast<uint32_t *>(lut)[lane] = reinterpret_cast<const uint32_t *>(dna4_lut)[lane];
    __syncthreads();
    for (uint32_t i = lane; i < len; i += 64) {
        code[i] = lut[bases[off + i]];
        qrank[i] = (uint8_t)(quals[off + i] - 33u);
    }
    __syncthreads();
    const uint32_t nk = len >= k ? len - k + 1 : 0;
    if (nk == 0) {                                   // window shorter than k: nothing to sample
        for (uint32_t s = lane; s < p; s += 64) {
            out_hash[(size_t)w * p + s] = 0;
            out_pos[(size_t)w * p + s] = 0;
        }
        if (lane == 0) out_has[w] = 0;
        return;
    }
    // indices of the k-mers whose quality sum reaches the threshold (:325-327), ascending
    uint32_t n_good = 0;
    for (uint32_t base = 0; base < nk; base += 64) {
        const uint32_t j = base + lane;
        bool ok = false;
        if (j < nk) {
            uint32_t qs = 0;
            for (uint32_t t = 0; t < k; t++) qs += qrank[j + t];
            ok = qs >= minq;
        }
        const uint64_t m = __ballot(ok);
        if (ok) good[n_good + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)j;
        n_good += (uint32_t)__popcll(m);
    }
    __syncthreads();
    const bool all = n_good == 0;                    // none is high-quality: consider all of them (:330-332)
    const uint32_t n_sel = all ? nk : n_good;
    for (uint32_t s = lane; s < p; s += 64) {        // Sampler(p) over the selection (:333-335), tabulated in fp64
        const uint32_t at = pos_table[(size_t)n_sel * p + s];
        const uint32_t j = all ? at : good[at];
        uint32_t h = 0;
        for (uint32_t t = 0; t < k; t++) h = (h << 2) | code[j + t];
        out_hash[(size_t)w * p + s] = h;
        out_pos[(size_t)w * p + s] = (uint16_t)j;
    }
    if (lane == 0) out_has[w] = 1;
}


}  // namespace bml
