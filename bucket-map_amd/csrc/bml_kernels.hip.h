// bml_kernels.hip.h -- gfx950 kernels of the locator candidate scan (include/bml.h).
//
//   bml_scan_kernel   : one 1 024-thread workgroup per (bucket, chunk of <= max_pairs candidates).  Packs the
//                       bucket to 2 bits/base in LDS and puts the DISTINCT k-mers the chunk's candidates ask for into
//                       an LDS open-addressing table (+ a presence filter).  The work the reference does by building an
//                       unordered_multimap of ALL k-mers per bucket (bucket_locator.h:162-177) and calling equal_range
//                       per sample (:246) is a join of the bucket's k-mers with the chunk's targets; its size (the
//                       occurrences) is unbounded in repeats -- a sampled k-mer of a satellite occurs 1 500 times in a
//                       bucket and a hundred candidates ask for it -- so nothing here is done once per occurrence
//                       except the one coalesced store that writes it:
//                         sweep 1  every k-mer of the bucket: filter, table; COUNT per distinct k-mer (O(bucket));
//                         layout   a target (candidate, sample) has as many occurrences as its k-mer was counted:
//                                  exclusive prefix over the <= 2 048 targets = every candidate's count and segment,
//                                  one atomic add reserves the chunk's stretch of the occurrence buffer;
//                         sweep 2  every k-mer of the bucket again: its rank among its equals, and ONE store -- into the
//                                  list of the first target that asked for it (O(bucket));
//                         copy     every other target with that k-mer copies the list (a wave per list, 512-byte
//                                  stores; a workgroup for long lists) and puts its own id in the keys.
//                       A candidate's segment comes out grouped by sample, in processing order (inside a group: any
//                       order).  No counting per occurrence, no second walk of the table per occurrence, no LDS staging.
//   bml_replay_light_kernel : one thread per candidate with at most kLightMax occurrences: orders them by
//                       (sample in processing order, offset descending) -- the order libstdc++'s equal_range
//                       yields equal keys -- and replays _find_offset's order-dependent vote
//                       (bucket_locator.h:233-288); its std::map<int,unsigned> is a sorted array, all in the thread's
//                       LDS slice.
//   bml_replay_heavy_kernel : one workgroup per candidate with more occurrences (a k-mer of a tandem repeat, a
//                       satellite or a poly-A stretch occurs thousands of times in a bucket, and a sorted array
//                       costs O(n^2) then): the same vote over DENSE bitmaps of the start positions, which needs no
//                       sorting at all -- see the kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bm_dna4.hip.h"

// threads of a heavy-replay workgroup (one candidate each).  Measured on the genome-like 1 M-read batch (builds with
// -DBML_HEAVY_THREADS=n): 64: 25.4 ms, 128: 18.8, 256: 15.0, 512: 21.1, 1024: 48.1 -- LDS holds six candidates a CU whatever
// the size, fewer threads leave the CU short of waves, more make every barrier dearer.
#ifndef BML_HEAVY_THREADS
#define BML_HEAVY_THREADS 256
#endif
namespace bml {

constexpr int kThreads = BML_HEAVY_THREADS;        // replay kernel
constexpr int kScanThreads = 1024;   // scan kernel: its loops are chains of dependent LDS reads, and the 66 KB of
                                     // LDS per workgroup allow two workgroups per CU -- 16 waves each hide the latency
                                     // that 4 waves each did not (6.96 -> 2.70 ms per 1 M candidates)
constexpr uint32_t kTableSlots = 4096;        // LDS open-addressing table: the distinct k-mers one chunk asks for
constexpr uint32_t kMaxTargets = kTableSlots / 2;   // (candidate, sample) pairs per chunk: the table stays at most half full
constexpr uint32_t kSpecialSlot = kTableSlots;      // slot of the one hash that equals kEmpty (k = 16: TTTTTTTTTTTTTTTT)
constexpr uint32_t kTableAlloc = kTableSlots + 4;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kFilterWords = 2048;       // 64 Kbit presence filter in front of the table: a bucket has 65 825 k-mers
                                              // and a chunk asks for at most 2 040 of all 4^k, so 97 % of the scan's
                                              // probes end at ONE LDS read instead of walking a half-full table
constexpr uint32_t kLongList = 1024;          // occurrence lists from this length on are copied by the whole workgroup
constexpr uint32_t kStage = 896;              // asked-for k-mers of the bucket that sweep 1 remembers (slot << 20 | offset): a chunk
                                              // outside repeats has a few hundred, and its sweep 2 is a walk of this list

struct Chunk {
    uint32_t bucket;
    uint32_t pair_begin;
    uint32_t pair_count;
};

struct LocParams {
    uint32_t k, p;
    int32_t allowed_mismatch, allowed_indel;
    uint32_t max_words;        // LDS words reserved for the packed bucket
    uint32_t max_pairs;        // most candidates a chunk holds: max_pairs * p <= kMaxTargets
};

// LDS layout of the scan (all dynamic, 16-byte aligned base):
//   header 128 B: gbase u64 | total u32 | n_short u32 | n_long u32 | n_stage u32 | ... | wsum[16] u32 at byte 64
//   filter[kFilterWords] u32          (after sweep 2: short_list u16[kMaxTargets] | long_list u16[kMaxTargets])
//   tkey[kTableAlloc] u32             the distinct k-mers (kEmpty = free)
//   tcnt[kTableAlloc] u32             occurrences of each in the bucket (sweep 2 counts it down again)
//   thead[kTableAlloc] u16            the first target that asked for it: the owner of the list
//   tstart[kMaxTargets + 4] u32       exclusive prefix of the targets' occurrence counts (relative to gbase)
//   tslot[kMaxTargets] u16            every target's table slot
//   stage[kStage] u32                 the asked-for k-mers sweep 1 met, while they fit
//   packed[max_words + 4] u32
__host__ __device__ inline size_t scan_lds_bytes(uint32_t max_words) {
    return 128 + (size_t)kFilterWords * 4 + (size_t)kTableAlloc * (4 + 4 + 2) + (size_t)(kMaxTargets + 4) * 4 + (size_t)kMaxTargets * 2 +
           (size_t)kStage * 4 + ((size_t)max_words + 4) * 4;
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

// occ_count[0]: occurrences reserved so far; occ_count[1]: set when a `placed` scan's recount disagrees with the segments
// it was given (never expected: the host checks it after every scan).
__global__ __launch_bounds__(kScanThreads) void bml_scan_kernel(
    LocParams P, const uint8_t *__restrict__ genome, const uint64_t *__restrict__ bucket_start,
    const uint32_t *__restrict__ bucket_len, const Chunk *__restrict__ chunks,
    const uint32_t *__restrict__ sample_hash, const uint32_t *__restrict__ pair_window,
    const uint8_t *__restrict__ pair_rc, uint32_t *__restrict__ occ, unsigned long long *__restrict__ occ_count,
    unsigned long long occ_cap, uint64_t *__restrict__ cand_start, uint32_t *__restrict__ cand_count,
    uint32_t *__restrict__ samp_end, uint32_t placed) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    unsigned long long &gbase = *reinterpret_cast<unsigned long long *>(smem);
    uint32_t &total = *reinterpret_cast<uint32_t *>(smem + 8);
    uint32_t &n_short = *reinterpret_cast<uint32_t *>(smem + 12);
    uint32_t &n_long = *reinterpret_cast<uint32_t *>(smem + 16);
    uint32_t *wsum = reinterpret_cast<uint32_t *>(smem + 64);
    uint32_t *filter = reinterpret_cast<uint32_t *>(smem + 128);
    uint32_t *tkey = filter + kFilterWords;
    uint32_t *tcnt = tkey + kTableAlloc;
    uint16_t *thead = reinterpret_cast<uint16_t *>(tcnt + kTableAlloc);
    uint32_t *tstart = reinterpret_cast<uint32_t *>(thead + kTableAlloc);
    uint16_t *tslot = reinterpret_cast<uint16_t *>(tstart + kMaxTargets + 4);
    uint32_t *stage = reinterpret_cast<uint32_t *>(tslot + kMaxTargets);
    uint32_t *packed = stage + kStage;
    uint32_t &n_stage = *reinterpret_cast<uint32_t *>(smem + 20);
    uint16_t *short_list = reinterpret_cast<uint16_t *>(filter), *long_list = short_list + kMaxTargets;   // (after sweep 2)

    const Chunk ch = chunks[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nb = bucket_len[ch.bucket];
    const uint64_t start = bucket_start[ch.bucket];
    const uint32_t n_t = ch.pair_count * P.p;                    // <= kMaxTargets (the host cuts the chunks so)

    if (tid == 0) n_short = n_long = n_stage = 0;
    for (uint32_t s = tid; s < kTableAlloc; s += kScanThreads) {
        tkey[s] = kEmpty;
        tcnt[s] = 0;
    }
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
    // the k-mers this chunk's candidates ask for: target t = (candidate, i-th processed sample); equal k-mers share a slot
    for (uint32_t t = tid; t < n_t; t += kScanThreads) {
        const uint32_t pair = ch.pair_begin + t / P.p, i = t % P.p;
        const uint32_t w = pair_window[pair];
        const bool rc = pair_rc[pair] != 0;
        // bucket_locator.h:235-242: reverse-complement candidates start from the last sample
        uint32_t h = sample_hash[(size_t)w * P.p + (rc ? P.p - 1u - i : i)];
        if (rc) h = hash_reverse_complement(h, P.k);
        uint32_t slot;
        if (h == kEmpty) {                                       // cannot be told from a free slot: a slot of its own
            slot = kSpecialSlot;
            if (atomicCAS(&tkey[slot], kEmpty, 0u) == kEmpty) thead[slot] = (uint16_t)t;
        } else {
            slot = slot_of(h);
            for (;;) {
                const uint32_t old = atomicCAS(&tkey[slot], kEmpty, h);
                if (old == kEmpty) thead[slot] = (uint16_t)t;    // the first to ask for it owns the list
                if (old == kEmpty || old == h) break;
                slot = (slot + 1u) & (kTableSlots - 1u);
            }
        }
        tslot[t] = (uint16_t)slot;
        const uint32_t fb = filter_bit(h);
        atomicOr(&filter[fb >> 5], 1u << (fb & 31u));
    }
    __syncthreads();

    // every k-mer of the bucket (bucket_locator.h:172-176 enumerates the same k-mers): hit(slot, j) for every k-mer j
    // that somebody asked for
    const uint32_t nk = nb >= P.k ? nb - P.k + 1u : 0u;
    const uint32_t kmask = P.k >= 16 ? 0xFFFFFFFFu : ((1u << (2u * P.k)) - 1u);
    // A thread takes the 16 k-mers that START in one word of the 2-bit stream: two LDS reads (its word and the next -- the
    // lanes of a wave read consecutive words) and one 64-bit shift per k-mer, instead of two reads and the address arithmetic
    // per k-mer (the scan on bases without repeats is bound by its VALU instruction count: profiles/r04/locate_sq_counters_*).
    const uint32_t top = 64u - 2u * P.k;                          // a k-mer that starts at the word's first base: shift by `top`
    const uint32_t n_groups = (shift + nk + 15u) >> 4;
    auto sweep = [&](auto hit) {
        for (uint32_t g = tid; g < n_groups; g += kScanThreads) {
            const uint64_t two = ((uint64_t)packed[g] << 32) | packed[g + 1u];
            const uint32_t j0 = 16u * g - shift;                  // bucket offset of the word's first base (wraps below 0: caught by j < nk)
#pragma unroll
            for (uint32_t o = 0; o < 16u; o++) {
                const uint32_t j = j0 + o;
                if (j >= nk) continue;                            // before the bucket's first base, or no whole k-mer left
                const uint32_t h = (uint32_t)(two >> (top - 2u * o)) & kmask;
                const uint32_t fb = filter_bit(h);
                if (!((filter[fb >> 5] >> (fb & 31u)) & 1u)) continue;      // nobody asked for this k-mer
                uint32_t slot;
                if (h == kEmpty) {
                    slot = tkey[kSpecialSlot] != kEmpty ? kSpecialSlot : kNone;
                } else {
                    slot = slot_of(h);
                    for (;;) {
                        const uint32_t key = tkey[slot];
                        if (key == h) break;
                        if (key == kEmpty) {
                            slot = kNone;
                            break;
                        }
                        slot = (slot + 1u) & (kTableSlots - 1u);
                    }
                }
                if (slot != kNone) hit(slot, j);
            }
        }
    };
    sweep([&](uint32_t slot, uint32_t j) {
        atomicAdd(&tcnt[slot], 1u);
        // remembered for sweep 2 while the list has room (one LDS atomic per wave and step, not per k-mer)
        const uint64_t m = __ballot(1), special = __ballot(slot == kSpecialSlot);   // (the special slot does not fit an entry:
        const int leader = __builtin_ctzll(m);                                      //  its k-mers make the list overflow)
        uint32_t at = 0;
        if ((int)lane == leader) at = atomicAdd(&n_stage, (uint32_t)__popcll(m) + (special ? kStage + 1u : 0u));
        at = (uint32_t)__shfl((int)at, leader, 64) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (at < kStage && slot != kSpecialSlot) stage[at] = (slot << 20) | j;
    });
    __syncthreads();

    // layout: tstart = exclusive prefix of the targets' counts (two targets a thread, a wave scan, the 16 wave totals)
    {
        const uint32_t t0 = 2u * tid, t1 = t0 + 1u;
        const uint32_t v0 = t0 < n_t ? tcnt[tslot[t0]] : 0u, v1 = t1 < n_t ? tcnt[tslot[t1]] : 0u;
        uint32_t incl = v0 + v1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += u;
        }
        if (lane == 63u) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += wsum[w];
        const uint32_t excl = before + incl - (v0 + v1);
        tstart[t0] = excl;
        tstart[t1] = excl + v0;
        if (tid == kScanThreads - 1u) {
            tstart[kMaxTargets] = before + incl;
            total = before + incl;
        }
    }
    __syncthreads();
    // one stretch of the occurrence buffer for the chunk, one segment in it per candidate (a candidate belongs to ONE chunk)
    if (tid == 0) gbase = placed ? (unsigned long long)cand_start[ch.pair_begin] : atomicAdd(occ_count, (unsigned long long)total);
    __syncthreads();
    const unsigned long long base = gbase;
    for (uint32_t c = tid; c < ch.pair_count; c += kScanThreads) {
        const uint32_t first = tstart[c * P.p], count = tstart[(c + 1u) * P.p] - first;
        if (placed) {
            // the scan before this one counted the same occurrences and the host placed the segments from its counts
            if (cand_count[ch.pair_begin + c] != count || cand_start[ch.pair_begin + c] != base + first) occ_count[1] = 1ull;
        } else {
            cand_count[ch.pair_begin + c] = count;
            cand_start[ch.pair_begin + c] = base + first;
        }
    }
    // where every sample's occurrences end inside its candidate's segment (the replay kernels' group boundaries)
    for (uint32_t t = tid; t < n_t; t += kScanThreads) samp_end[(size_t)ch.pair_begin * P.p + t] = tstart[t + 1u] - tstart[(t / P.p) * P.p];
    if (total == 0u || base + total > occ_cap) return;                     // (too small: the host grows the buffer and scans again)

    // sweep 2: the rank of every asked-for k-mer among its equals, and one store into its owner's list
    auto place = [&](uint32_t slot, uint32_t j) {
        const uint32_t rank = atomicSub(&tcnt[slot], 1u) - 1u;
        occ[base + tstart[thead[slot]] + rank] = j;                         // an occurrence is its offset in the bucket: 4 bytes
    };
    const uint32_t staged = n_stage;
    if (staged <= kStage) {                                                // (never when the special slot was met)
        for (uint32_t e = tid; e < staged; e += kScanThreads) place(stage[e] >> 20, stage[e] & 0xFFFFFu);
    } else {
        sweep(place);
    }
    // The lists are read back below by this workgroup only: __syncthreads() orders them at workgroup scope (all its waves
    // share the CU's vector cache).  A device-scope fence here writes the XCD's whole L2 back, once per workgroup: 11 ms
    // instead of 1.5 per million candidates.
    __syncthreads();
    // the targets that asked for a k-mer somebody else owns: their lists are copies
    for (uint32_t t = tid; t < kMaxTargets; t += kScanThreads) {
        uint32_t len = 0;
        if (t < n_t && thead[tslot[t]] != t) len = tstart[t + 1u] - tstart[t];
        const uint64_t ms = __ballot(len != 0u && len < kLongList), ml = __ballot(len >= kLongList);
        uint32_t bs = 0, bl = 0;
        if (lane == 0) {
            if (ms) bs = atomicAdd(&n_short, (uint32_t)__popcll(ms));
            if (ml) bl = atomicAdd(&n_long, (uint32_t)__popcll(ml));
        }
        bs = (uint32_t)__shfl((int)bs, 0, 64);
        bl = (uint32_t)__shfl((int)bl, 0, 64);
        const uint64_t below = (1ull << lane) - 1ull;
        if (len != 0u && len < kLongList) short_list[bs + (uint32_t)__popcll(ms & below)] = (uint16_t)t;
        if (len >= kLongList) long_list[bl + (uint32_t)__popcll(ml & below)] = (uint16_t)t;
    }
    __syncthreads();
    auto copy_list = [&](uint32_t t, uint32_t first, uint32_t step) {
        const uint32_t owner = thead[tslot[t]], len = tstart[t + 1u] - tstart[t];
        const uint32_t *src = occ + base + tstart[owner];
        uint32_t *dst = occ + base + tstart[t];
        for (uint32_t e = first; e < len; e += step) dst[e] = src[e];
    };
    const uint32_t ns = n_short, nl = n_long;
    for (uint32_t i = 0; i < nl; i++) copy_list(long_list[i], tid, kScanThreads);
    for (uint32_t i = wave; i < ns; i += kScanThreads / 64u) copy_list(short_list[i], lane, 64u);
}

// Occurrences a candidate may have for the one-thread replay: room for one true occurrence per sample and a few chance
// ones -- 16 up to p = 10 samples, 32 up to p = 24, 64 beyond (the LDS slice grows with it: fewer threads per CU).

// _find_offset (bucket_locator.h:209-290) for one candidate per thread: candidates with at most kLightMax occurrences
// (the common case: one true occurrence per sample and a few chance ones).  The others go to `heavy`.
// Everything a thread touches more than once lives in its LDS slice -- its occurrences as 32-bit sort keys
// (sample << 20 | 0xFFFFF - offset: buckets are shorter than 2^20 bases, bml_create), the proposal map as a sorted
// array of starts with 8-bit votes -- laid out entry-major ([entry][thread]), so that the threads of a wave that are
// at the same entry hit 64 different banks.  9 bytes x 16 entries x 256 threads = 36 KB per workgroup.
template <int kLightMax, int kLightThreads>
__global__ __launch_bounds__(kLightThreads) void bml_replay_light_kernel(
    LocParams P, const uint32_t *__restrict__ occ, const uint64_t *__restrict__ cand_start, const uint32_t *__restrict__ cand_count,
    const uint32_t *__restrict__ samp_end, const uint16_t *__restrict__ sample_pos, const uint32_t *__restrict__ seg_len,
    const uint32_t *__restrict__ pair_window, const uint8_t *__restrict__ pair_rc, uint32_t pair_lo, uint32_t pair_hi,
    int32_t *__restrict__ out_offset, uint32_t *__restrict__ out_votes, uint32_t *__restrict__ heavy, uint32_t *__restrict__ n_heavy) {
    __shared__ uint32_t s_key[kLightMax][kLightThreads];
    __shared__ int32_t s_pk[kLightMax][kLightThreads];
    __shared__ uint8_t s_pv[kLightMax][kLightThreads];
    const uint32_t tid = threadIdx.x, pair = pair_lo + blockIdx.x * kLightThreads + tid;   // candidates [pair_lo, pair_hi)
    if (pair >= pair_hi) return;
    const uint32_t n = cand_count[pair];
    if (n > kLightMax) {
        heavy[atomicAdd(n_heavy, 1u)] = pair;
        atomicMax(n_heavy + 1, n);
        return;
    }
    // (sample in processing order) ascending, offset descending: the segment arrives grouped by sample; insertion sort while loading
    const uint32_t *mine = occ + cand_start[pair];
    const uint32_t *ends = samp_end + (size_t)pair * P.p;
    uint32_t sample = 0, sample_stop = ends[0];
    for (uint32_t a = 0; a < n; a++) {
        while (a >= sample_stop) sample_stop = ends[++sample];      // (a < n = ends[p - 1]: stays in range)
        const uint32_t v = (sample << 20) | (0xFFFFFu - mine[a]);
        uint32_t b = a;
        while (b > 0 && s_key[b - 1][tid] > v) {
            s_key[b][tid] = s_key[b - 1][tid];
            b--;
        }
        s_key[b][tid] = v;
    }
    uint32_t np = 0;
    const uint32_t w = pair_window[pair];
    const bool rc = pair_rc[pair] != 0;
    const uint32_t length = seg_len[w];
    uint32_t cur = 0;
    for (uint32_t i = 0; i < P.p && cur < n; i++) {
        if ((s_key[cur][tid] >> 20) != i) continue;               // no occurrence of this sample
        uint32_t idx = sample_pos[(size_t)w * P.p + (rc ? P.p - 1u - i : i)];
        if (rc) idx = length - P.k - idx;                        // :242 where the k-mer starts on the other strand
        const bool was_empty = np == 0;                          // :247 evaluated once per sample
        while (cur < n && (s_key[cur][tid] >> 20) == i) {
            const uint32_t occ = 0xFFFFFu - (s_key[cur][tid] & 0xFFFFFu);
            const int32_t position = (int32_t)(occ - idx);       // :250,257 (unsigned arithmetic, wraps like int)
            // lower_bound(position - indel) .. upper_bound(position + indel)  (:259-260); the map is small: linear
            uint32_t lb = 0, ub = 0;
            if (!was_empty) {
                const int32_t lo_key = position - P.allowed_indel, hi_key = position + P.allowed_indel;
                while (lb < np && s_pk[lb][tid] < lo_key) lb++;
                ub = lb;
                while (ub < np && s_pk[ub][tid] <= hi_key) ub++;
            }
            if (!was_empty && lb < ub) {
                for (uint32_t v = lb; v < ub; v++) s_pv[v][tid]++;     // :262-265 every proposal in range gets the vote
            } else {
                // vote_counter[position]++ (:251,269): insert, or increment when the key exists
                uint32_t a = 0;
                while (a < np && s_pk[a][tid] < position) a++;
                if (a < np && s_pk[a][tid] == position) {
                    s_pv[a][tid]++;
                } else {
                    for (uint32_t v = np; v > a; v--) {
                        s_pk[v][tid] = s_pk[v - 1][tid];
                        s_pv[v][tid] = s_pv[v - 1][tid];
                    }
                    s_pk[a][tid] = position;
                    s_pv[a][tid] = 1;
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
            if (s_pv[v][tid] > s_pv[best][tid]) best = v;
        // :284 unsigned >= int compares as unsigned
        if ((uint32_t)s_pv[best][tid] >= (uint32_t)((int32_t)P.p - P.allowed_mismatch) && s_pk[best][tid] >= 0) {
            off = s_pk[best][tid];
            votes = s_pv[best][tid];
        }
    }
    out_offset[pair] = off;
    out_votes[pair] = votes;
}

// _find_offset for a candidate with MANY occurrences, one workgroup per candidate, without sorting anything.
//
// The reference keeps its proposals in a std::map and, for every occurrence of sample i in descending offset order:
// if the map was empty when the sample began the start position is inserted; else every proposal within +-indel of
// the start gets a vote, or -- if there is none -- the start is inserted with one vote (:247-270).  Start positions
// lie in [-window length, bucket length): a DENSE bitmap `exists` of that range and a LIST of the proposals (start, votes)
// replace the map, and the sample's occurrences are handled in order-free steps:
//   1. the sample's starts are set in a bitmap `starts`; a start with no proposal within +-indel is marked in `fresh`.
//   2. the proposals that existed BEFORE the sample (K): a thread per proposal adds the number of starts within +-indel of
//      it -- a population count over `starts`.  Those are the votes the reference gives one (occurrence, proposal) pair
//      at a time; as atomic adds on a dense vote array they were 1.2 G scattered atomics per million reads in repeats and a
//      third of the kernel's time.
//   3. the starts in `fresh`, highest first (= the reference's processing order): one becomes a new proposal iff
//      the proposal inserted last lies more than indel above it (proposals made earlier in this sample are all above
//      it, the last one is the nearest; those of K are not in range by step 1).  One wave walks the bitmap.
//   4. a new proposal begins with its own vote and gets one from every start within indel BELOW it (new proposals are more
//      than indel apart, so a start votes for one at most; the proposal was inserted before those starts were processed,
//      and starts above it were processed before it existed); then it joins `exists`.
// Winner: most votes, ties -> smallest start (:281-283), accepted as in the light kernel.
// Bitmaps live in LDS when 3 of them fit (`lds_bitmaps`), else in the workgroup's global scratch; nothing else does.

struct HeavyScratch {
    uint32_t *votes;        // per workgroup: 2 x vote_stride entries (the proposals' start positions, then their votes)
    uint32_t *bitmaps;      // per workgroup: 3 x words (used when the bitmaps do not fit LDS)
    uint32_t vote_stride;   // min(range, the most occurrences a heavy candidate has): a candidate makes at most one proposal
                            // per start position and at most one per occurrence
};

__global__ __launch_bounds__(kThreads) void bml_replay_heavy_kernel(
    LocParams P, const uint32_t *__restrict__ occ, const uint64_t *__restrict__ cand_start,
    const uint32_t *__restrict__ cand_count, const uint32_t *__restrict__ samp_end, const uint16_t *__restrict__ sample_pos,
    const uint32_t *__restrict__ seg_len,
    const uint32_t *__restrict__ pair_window,
    const uint8_t *__restrict__ pair_rc, const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ n_heavy,
    uint32_t range, uint32_t bias, uint32_t lds_bitmaps, HeavyScratch S, int32_t *__restrict__ out_offset,
    uint32_t *__restrict__ out_votes) {
    extern __shared__ uint32_t heavy_lds[];
    __shared__ uint32_t s_hist[65], s_fresh, s_new, s_np, s_fmax, s_fmin, s_next;
    __shared__ unsigned long long s_best;
    const uint32_t tid = threadIdx.x, words = (range + 31u) / 32u;
    uint32_t *exists = lds_bitmaps ? heavy_lds : S.bitmaps + (size_t)blockIdx.x * 3u * words;
    uint32_t *fresh = exists + words, *starts = fresh + words;
    // the proposals, in the order they were made: start position and votes (at most one per start position)
    uint32_t *prop_pos = S.votes + (size_t)blockIdx.x * 2u * S.vote_stride, *prop_votes = prop_pos + S.vote_stride;
    const int32_t d = P.allowed_indel;
    const uint32_t total = *n_heavy;
    // The bitmaps are all zero between candidates: a candidate only ever sets bits at the start positions of its own
    // occurrences, and clears exactly those words when it is done -- per candidate the work is O(occurrences), not O(range).
    for (uint32_t i = tid; i < 3u * words; i += kThreads) exists[i] = 0;
    __syncthreads();
    // bm's words over [lo, hi] (inclusive, clamped to the range), masked to it: fn(bits) for each; false stops the walk
    auto over_bits = [&](const uint32_t *bm, int64_t lo, int64_t hi, auto fn) {
        if (lo < 0) lo = 0;
        if (hi >= (int64_t)range) hi = (int64_t)range - 1;
        if (lo > hi) return;
        for (uint32_t wd = (uint32_t)lo >> 5; wd <= ((uint32_t)hi >> 5); wd++) {
            uint32_t bits = bm[wd];
            if (wd == ((uint32_t)lo >> 5)) bits &= 0xFFFFFFFFu << ((uint32_t)lo & 31u);
            if (wd == ((uint32_t)hi >> 5)) bits &= 0xFFFFFFFFu >> (31u - ((uint32_t)hi & 31u));
            if (!fn(bits)) return;
        }
    };
    auto any_bit = [&](const uint32_t *bm, int64_t lo, int64_t hi) {
        bool any = false;
        over_bits(bm, lo, hi, [&](uint32_t bits) {
            any = bits != 0;
            return !any;
        });
        return any;
    };
    auto count_bits = [&](const uint32_t *bm, int64_t lo, int64_t hi) {
        uint32_t count = 0;
        over_bits(bm, lo, hi, [&](uint32_t bits) {
            count += (uint32_t)__builtin_popcount(bits);
            return true;
        });
        return count;
    };
    // candidates are handed out one at a time (their costs differ a hundredfold): n_heavy[2] is the queue's head
    for (;;) {
        __syncthreads();
        if (tid == 0) s_next = atomicAdd(const_cast<uint32_t *>(n_heavy) + 2, 1u);
        __syncthreads();
        const uint32_t h = s_next;
        if (h >= total) break;
        const uint32_t pair = heavy[h];
        const uint64_t first = cand_start[pair];
        const uint32_t w = pair_window[pair];
        const bool rc = pair_rc[pair] != 0;
        const uint32_t length = seg_len[w];
        // 0. the scan wrote the candidate's occurrences grouped by sample, in processing order, and where every group ends
        const uint32_t *mine = occ + first;
        if (tid == 0) {
            s_np = 0;
            s_hist[0] = 0;
        }
        if (tid < P.p) s_hist[tid + 1] = samp_end[(size_t)pair * P.p + tid];
        __syncthreads();
        for (uint32_t i = 0; i < P.p; i++) {
            uint32_t idx = sample_pos[(size_t)w * P.p + (rc ? P.p - 1u - i : i)];
            if (rc) idx = length - P.k - idx;
            const uint32_t o0 = s_hist[i], o1 = s_hist[i + 1];
            if (o0 == o1) continue;                                  // (uniform: s_hist is shared)
            // start position of an occurrence, shifted by `bias` into [0, range)
            auto start_of = [&](uint32_t occ) { return (uint32_t)((int32_t)(occ - idx) + (int32_t)bias); };
            // every step below walks the sample's occurrences again (global memory: four loads in flight, one latency)
            auto for_each_pos = [&](auto fn) {
                for (uint32_t o = o0 + tid; o < o1; o += 4u * kThreads) {
                    uint32_t at[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) at[u] = o + u * kThreads < o1 ? mine[o + u * kThreads] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (o + u * kThreads < o1) fn(start_of(at[u]), o + u * kThreads - o0);
                }
            };
            const uint32_t np = s_np;                                // proposals before this sample (uniform)
            if (np == 0) {                                           // :247-251 the map was empty: every start goes in
                __syncthreads();
                for_each_pos([&](uint32_t pos, uint32_t nth) {
                    atomicOr(&exists[pos >> 5], 1u << (pos & 31u));
                    prop_pos[nth] = pos;
                    prop_votes[nth] = 1;
                });
                if (tid == 0) s_np = o1 - o0;
                __syncthreads();
                continue;
            }
            if (tid == 0) {
                s_fresh = s_new = 0;
                s_fmax = 0;
                s_fmin = 0xFFFFFFFFu;
            }
            __syncthreads();
            // 1. the sample's starts as a bitmap; a start with no proposal within +-indel is marked in `fresh`
            for_each_pos([&](uint32_t pos, uint32_t) {
                atomicOr(&starts[pos >> 5], 1u << (pos & 31u));
                if (!any_bit(exists, (int64_t)pos - d, (int64_t)pos + d)) {
                    atomicOr(&fresh[pos >> 5], 1u << (pos & 31u));
                    atomicAdd(&s_fresh, 1u);
                    atomicMax(&s_fmax, pos);
                    atomicMin(&s_fmin, pos);
                }
            });
            __syncthreads();
            // 2. votes for the proposals that were there before this sample: each counts the starts within +-indel of it
            //    (a thread per proposal and a population count -- not an atomic add per (occurrence, proposal) pair)
            //    (four proposals a thread at a time: the list lives in global memory, and four loads in flight cost one latency)
            for (uint32_t q0 = tid; q0 < np; q0 += 4u * kThreads) {
                uint32_t at[4], got[4];
#pragma unroll
                for (int u = 0; u < 4; u++) at[u] = q0 + u * kThreads < np ? prop_pos[q0 + u * kThreads] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    got[u] = q0 + u * kThreads < np ? count_bits(starts, (int64_t)at[u] - d, (int64_t)at[u] + d) : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (got[u]) atomicAdd(&prop_votes[q0 + u * kThreads], got[u]);   // (no return value: fire and forget)
            }
            // 3. new proposals among the fresh starts, highest first (one wave)
            if (s_fresh != 0 && tid < 64) {
                int64_t limit = (int64_t)s_fmax;                     // the next proposal lies at or below `limit`
                const int64_t lowest = (int64_t)s_fmin;
                uint32_t made = 0;
                while (limit >= lowest) {
                    // the 64 words at and below limit's word, lane l takes word (top - l)
                    const int64_t top = limit >> 5, wd = top - (int64_t)tid;
                    uint32_t bits = wd >= 0 ? fresh[wd] : 0u;
                    if (tid == 0) bits &= 0xFFFFFFFFu >> (31u - ((uint32_t)limit & 31u));
                    const uint64_t m = __ballot(bits != 0);
                    if (m == 0) {
                        limit = (top - 64) * 32 + 31;
                        continue;
                    }
                    const int src = __builtin_ctzll(m);
                    const uint32_t hb = 31u - (uint32_t)__builtin_clz((uint32_t)__shfl((int)bits, src, 64));
                    const uint32_t pos = (uint32_t)(top - src) * 32u + hb;
                    if (tid == 0) prop_pos[np + made] = pos;
                    made++;
                    limit = (int64_t)pos - d - 1;                    // nothing within indel below a new proposal
                }
                if (tid == 0) s_new = made;
            }
            __syncthreads();
            // 4. a new proposal starts with its own vote and gets one from every start within indel BELOW it (those were
            //    processed after it was inserted; new proposals are more than indel apart, so a start votes for one at most);
            //    it joins the map
            for (uint32_t q = np + tid; q < np + s_new; q += kThreads) {
                const uint32_t at = prop_pos[q];
                prop_votes[q] = 1u + count_bits(starts, (int64_t)at - d, (int64_t)at - 1);
                atomicOr(&exists[at >> 5], 1u << (at & 31u));
            }
            __syncthreads();
            if (o1 - o0 >= words / 16u) {
                // a sample with more than a sixteenth as many occurrences as the bitmaps have words: sweeping both bitmaps is cheaper than
                // another pass over its occurrences in global memory (fresh follows starts in memory)
                for (uint32_t i = tid; i < 2u * words; i += kThreads) fresh[i] = 0;
            } else {
                for_each_pos([&](uint32_t pos, uint32_t) {           // (this sample's words only)
                    starts[pos >> 5] = 0;
                    fresh[pos >> 5] = 0;
                });
            }
            if (tid == 0) s_np = np + s_new;
            __syncthreads();
        }
        // winner: most votes, ties -> smallest start (:281-283)
        if (tid == 0) s_best = 0;
        __syncthreads();
        unsigned long long best = 0;
        for (uint32_t q0 = tid; q0 < s_np; q0 += 4u * kThreads) {
            uint32_t v[4], at[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool in = q0 + u * kThreads < s_np;
                v[u] = in ? prop_votes[q0 + u * kThreads] : 0u;
                at[u] = in ? prop_pos[q0 + u * kThreads] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned long long cand = ((unsigned long long)v[u] << 32) | (0xFFFFFFFFu - at[u]);
                best = cand > best ? cand : best;
            }
        }
        atomicMax(&s_best, best);
        __syncthreads();
        if (tid == 0) {
            int32_t off = -1;
            uint32_t nv = 0;
            if (s_best) {
                const uint32_t v = (uint32_t)(s_best >> 32);
                const int32_t key = (int32_t)(0xFFFFFFFFu - (uint32_t)s_best) - (int32_t)bias;
                if (v >= (uint32_t)((int32_t)P.p - P.allowed_mismatch) && key >= 0) {
                    off = key;
                    nv = v;
                }
            }
            out_offset[pair] = off;
            out_votes[pair] = nv;
        }
        for (uint32_t q0 = tid; q0 < s_np; q0 += 4u * kThreads) {  // leave the map empty for the next candidate
            uint32_t at[4];
#pragma unroll
            for (int u = 0; u < 4; u++) at[u] = q0 + u * kThreads < s_np ? prop_pos[q0 + u * kThreads] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (at[u] != 0xFFFFFFFFu) exists[at[u] >> 5] = 0;
        }
        __syncthreads();
    }
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
