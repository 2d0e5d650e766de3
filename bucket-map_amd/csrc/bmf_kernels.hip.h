// bmf_kernels.hip.h -- gfx950 kernels of the candidate-bucket filter.
//
// Two kernels per batch of read windows:
//   bmf_sample_kernel : one wave per window (persistent workgroups).  k-mer hashes, quality window sums, distinguishability +
//                       quality filter, ordered compaction, deterministic sampling, q-gram -> row-id
//                       lists for both orientations (reference: q_gram_mapper.h:431-469).
//   bmf_vote_kernel   : one wave per (window, orientation).  Streams the S*G index rows of the list
//                       through a DEPTH-deep register ring of 16-byte-per-lane loads, ANDs the G rows
//                       of each sample, keeps the per-bucket miss count bit-sliced in VGPRs, then
//                       finds the minimum and emits the ascending bucket ids
//                       (reference: q_gram_mapper.h:380-412 + fault_tolerate_filter :59-102).
//
// Formulation (differs from the reference on purpose; SURVEY.md 3.3 proves equivalence): the
// reference keeps F unary bit-planes lvl[i][b] == (misses(b) <= F-1-i).  Here misses(b) is kept as a
// PLANES-bit binary counter per bucket, bit-sliced across PLANES registers and saturating at
// 2^PLANES-1 >= F, so the result "buckets with the fewest misses, if that minimum is < F" is read off
// with a most-significant-plane-first minimum search.  Integer / bit work only: no MFMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bm_dna4.hip.h"

namespace bmf {

constexpr int kWave = 64;

struct DevParams {
    uint32_t nb;         // NB
    uint32_t k, q, G;    // G = k-q+1 q-grams per k-mer
    uint32_t S, F;
    uint32_t qbits;      // 4^q - 1
    uint32_t minq;       // min_base_quality = b*k
    uint32_t min_good;   // smallest #good k-mers that is NOT rejected by `size < 0.2*S` (fp64, host)
    uint32_t max_cand;
    uint32_t read_len;
    uint32_t max_kmers;  // read_len - k + 1
    uint32_t list_len;   // row ids per (window, orientation): S*G rounded up to the ring depth, + depth (padding)
    uint32_t n_chunks;   // ceil(ceil(NB/8)/16) : 16-byte chunks per row that hold buckets
    uint32_t pitch;      // bytes between rows in HBM (multiple of 128)
    uint32_t ones_row;   // id of the all-ones row appended after the index (for un-indexed q-grams)
    uint32_t n_kmers;    // entries of kmer_to_index (4^q, or 0 when no .kmers_index was loaded)
    uint32_t early_exit; // BMF_FLAG_EARLY_EXIT: a pruning kernel variant is in use (informational)
    uint32_t pass1_rows; // two-pass variant: q-gram rows of each sample read at full width in pass 1 (1..G)
    uint32_t max_live;   // two-pass variant: live chunks (= lanes) an item may bring to the recount kernel (16 or 32)
    uint32_t item_base;  // two-pass variant: first (window, orientation) item of this launch (the batch goes out in slices)
    uint32_t row_order;  // entry i of a sample's G row ids is the row of q-gram (row_order >> 4i) & 15: a permutation of 0..G-1
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// utils.h:291-302
__device__ __forceinline__ uint32_t hash_reverse_complement(uint32_t h, uint32_t k) {
    uint32_t rc = 0;
    for (uint32_t i = 0; i < k; i++) {
        rc = (rc << 2) | ((~h) & 3u);
        h >>= 2;
    }
    return rc;
}

// --------------------------------------------------------------------------------------------------
// index preparation (mapper::load side)
// --------------------------------------------------------------------------------------------------

// Clears the bits >= NB of the last index byte of every row (the reference's _bitset_from_bytes,
// q_gram_mapper.h:238-248, never reads them) so popcounts and ANDs see exactly NB bits.
// .qgram rows (row_bytes apart, any alignment) -> their 128-byte-pitched slots.  A strided 2-D copy from the
// host takes ~7 us per row (1.9 s for 262 144 rows); one flat copy plus this kernel takes milliseconds.
__global__ void bmf_repitch_kernel(const uint8_t *__restrict__ packed, uint8_t *__restrict__ rows, uint64_t n_rows,
                                   uint32_t row_bytes, uint32_t pitch) {
    const uint32_t words = (row_bytes + 3u) / 4u;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * words) return;
    const uint64_t r = i / words;
    const uint32_t b = (uint32_t)(i % words) * 4u;
    const uint8_t *src = packed + r * row_bytes + b;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t t = 0; t < 4; t++)
        if (b + t < row_bytes) v |= (uint32_t)src[t] << (8 * t);
    *reinterpret_cast<uint32_t *>(rows + r * pitch + b) = v;
}

// The inverse, for bmf_index_download: pitched rows -> the packed .qgram layout, byte by byte (packed rows
// start at any alignment).
__global__ void bmf_pack_rows_kernel(const uint8_t *__restrict__ rows, uint8_t *__restrict__ packed, uint64_t n_rows,
                                     uint32_t row_bytes, uint32_t pitch) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * row_bytes) return;
    packed[i] = rows[(i / row_bytes) * pitch + i % row_bytes];
}

__global__ void bmf_sanitize_rows_kernel(uint8_t *rows, uint64_t n_rows, uint32_t pitch, uint32_t nb) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows || (nb & 7u) == 0) return;
    uint32_t last = ((nb + 7u) >> 3) - 1u;
    rows[r * pitch + last] &= (uint8_t)(0xFFu >> (8u - (nb & 7u)));
}

// Exact pruning, folded first pass (bmf_vote2.hip.h): row g of the index FOLDED by f (2 or 4) has bit i set iff any of
// the buckets f*i .. f*i+f-1 is set in row g.  One thread per output word (32 groups).
__device__ __forceinline__ uint32_t fold4_byte(uint32_t w) {     // 32 bits -> 8: bit i = OR of bits 4i .. 4i+3
    uint32_t t = w | (w >> 1);
    t = (t | (t >> 2)) & 0x11111111u;
    t = (t | (t >> 3)) & 0x03030303u;
    t = (t | (t >> 6)) & 0x000F000Fu;
    return (t | (t >> 12)) & 0xFFu;
}
__device__ __forceinline__ uint32_t spread4_byte(uint32_t b) {   // 8 bits -> 32: bit i goes to bits 4i .. 4i+3
    uint32_t x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x * 0xFu;
}
__device__ __forceinline__ uint32_t fold2_half(uint32_t w) {     // 32 bits -> 16: bit i = OR of bits 2i, 2i+1
    uint32_t t = (w | (w >> 1)) & 0x55555555u;
    t = (t | (t >> 1)) & 0x33333333u;
    t = (t | (t >> 2)) & 0x0F0F0F0Fu;
    t = (t | (t >> 4)) & 0x00FF00FFu;
    return (t | (t >> 8)) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t spread2_half(uint32_t b) {   // 16 bits -> 32: bit i goes to bits 2i, 2i+1
    uint32_t x = (b | (b << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x * 3u;
}
template <int FOLD>
__global__ void bmf_fold_kernel(const uint8_t *__restrict__ rows, uint64_t n_rows, uint32_t pitch,
                                uint8_t *__restrict__ folded, uint32_t pitch_f) {
    const uint32_t words_f = pitch_f >> 2, words = pitch >> 2;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * words_f) return;
    const uint64_t r = i / words_f;
    const uint32_t ow = (uint32_t)(i % words_f);
    const uint32_t *in = reinterpret_cast<const uint32_t *>(rows + r * pitch);
    uint32_t out = 0;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)FOLD; k++)
        if ((uint32_t)FOLD * ow + k < words) {
            const uint32_t w = in[(uint32_t)FOLD * ow + k];
            out |= FOLD == 4 ? fold4_byte(w) << (8u * k) : fold2_half(w) << (16u * k);
        }
    reinterpret_cast<uint32_t *>(folded + r * pitch_f)[ow] = out;
}

// distinguishability_filter::read (q_gram_mapper.h:171-187): zeros[row] = NB - popcount(row).
// One wave per row, coalesced dword reads.
__global__ __launch_bounds__(kWave) void bmf_zeros_kernel(const uint8_t *rows, uint64_t n_rows, uint32_t pitch,
                                                         uint32_t nb, uint32_t *zeros) {
    for (uint64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(rows + r * pitch);
        uint32_t n_dw = pitch >> 2, ones = 0;
        for (uint32_t i = threadIdx.x; i < n_dw; i += kWave) ones += __popc(row[i]);
        ones = wave_sum(ones);
        if (threadIdx.x == 0) zeros[r] = nb - ones;   // ones == 0 -> NB, as the reference's special case
    }
}

// Bitmap over all 4^q q-grams: bit g set iff the q-gram is indexed and its row has
// zeros >= threshold (the per-q-gram half of is_highly_distinguishable, q_gram_mapper.h:189-196).
__global__ void bmf_qgram_ok_kernel(const int32_t *k2i, uint64_t n_kmers, const uint32_t *zeros,
                                    uint32_t threshold, uint32_t *bitmap, uint64_t n_words) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32; b++) {
        uint64_t g = w * 32 + b;
        if (g < n_kmers) {
            int32_t idx = k2i[g];
            if (idx >= 0 && zeros[idx] >= threshold) bits |= 1u << b;
        }
    }
    bitmap[w] = bits;
}

// --------------------------------------------------------------------------------------------------
// sample kernel: q_gram_mapper::query_sequence up to (not including) the two query() calls
// --------------------------------------------------------------------------------------------------
//
// One wave per window, `waves_per_wg` waves per workgroup, workgroups persistent over the windows.
// History of what bounds it (profiles/r02).  Round 1's kernel spent 15 us per window in a chain of
// dependent round trips (five rounds of byte loads for the bases, five rounds of bitmap gathers from L2); with
// those gone it turned out to be bound by VALU issue -- 632 wave-instructions per window at 4 cycles each on a
// 16-lane SIMD is 1.0 of its 1.65 ms -- so the instruction stream is what this version cuts:
//   * bases and qualities arrive as one round of aligned 8-byte loads per lane (the window may start at any
//     byte: lanes load the aligned chunks that cover it, positions are counted from the first chunk);
//   * ASCII -> dna4 rank four bytes at a time: ((c >> 1) & 3) ^ (that >> 1) is right for A C G T in either case,
//     v_perm_b32 rebuilds the letters those ranks stand for, and only a word that is NOT its own rebuild (N,
//     IUPAC, anything else) takes the byte-wise exact folding (SeqAn3's);
//   * the bases are kept as a 2-bit big-endian stream (16 per word): a k-mer or q-gram hash is one 64-bit shift
//     over two words; the k-mer quality sums (quality_filter.h:611-631) are differences of a prefix-sum array;
//   * is_highly_distinguishable (q_gram_mapper.h:189-196) asks, for each of the G = k-q+1 q-grams of a k-mer,
//     one bit of a 4^q-bit map; consecutive k-mers share all but one of them, so the map (staged in LDS once per
//     workgroup when BITMAP_LDS) is read once per POSITION, the bits of a round of 64 positions are a wave ballot,
//     and "any of my G q-grams" is G-1 scalar shift-ORs of that ballot with the next round's;
//   * the reverse complement of a hash is a bit reversal and three masks, not a loop over k bases.
//
// LDS (dynamic): [bitmap, bitmap_words u32] then per wave: pk[pk_bytes / 4] u32 (packed bases of the aligned chunks
//                covering the window, + one pad word) | qsum[qsum_bytes / 4] u32 (exclusive prefix sums of the phred
//                ranks over the same chunks, + the total) | goodh[max_kmers] u32; wave_stride bytes in all
struct SampleGeom {
    uint32_t n_windows;
    uint32_t waves_per_wg;
    uint32_t bitmap_words;   // u32 words of the q-gram bitmap (staged in LDS when BITMAP_LDS)
    uint32_t pk_bytes;       // multiples of 16
    uint32_t qsum_bytes;
    uint32_t wave_stride;    // LDS bytes per wave
};

using bmdna::dna4_code;
using bmdna::dna4_pack4;

// utils.h:291-302 without the loop: reverse the bits, swap the two bits of every base back, complement
__device__ __forceinline__ uint32_t revcomp_fast(uint32_t h, uint32_t k) {
    const uint32_t br = __brev(h);
    const uint32_t sw = ((br >> 1) & 0x55555555u) | ((br & 0x55555555u) << 1);
    return (~sw) >> (32u - 2u * k);
}

template <bool BITMAP_LDS>
__global__ __launch_bounds__(1024, 8) void bmf_sample_kernel(   // 8 waves per SIMD (two workgroups per CU): at most 64 VGPRs
    DevParams P, SampleGeom Gm, const uint8_t *__restrict__ bases, const uint8_t *__restrict__ quals,
    const uint64_t *__restrict__ win_start, const uint32_t *__restrict__ win_len,
    const uint32_t *__restrict__ qgram_ok, const int32_t *__restrict__ k2i,
    const uint16_t *__restrict__ pos_table, uint32_t *__restrict__ row_lists,
    uint32_t *__restrict__ list_n, uint32_t *__restrict__ rows_anded) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *okmap_lds = reinterpret_cast<uint32_t *>(smem);
    uint32_t lds_bitmap_bytes = 0;
    if (BITMAP_LDS) {
        for (uint32_t i = threadIdx.x; i < Gm.bitmap_words; i += blockDim.x) okmap_lds[i] = qgram_ok[i];
        lds_bitmap_bytes = (Gm.bitmap_words * 4u + 15u) & ~15u;
        __syncthreads();                       // the only workgroup barrier: from here on the waves are on their own
    }
    uint32_t *pk = reinterpret_cast<uint32_t *>(smem + lds_bitmap_bytes + (size_t)wave * Gm.wave_stride);
    uint16_t *pk16 = reinterpret_cast<uint16_t *>(pk);
    uint32_t *qsum = pk + Gm.pk_bytes / 4;
    uint32_t *goodh = qsum + Gm.qsum_bytes / 4;
    const uint32_t kmask = P.k >= 16 ? 0xFFFFFFFFu : (1u << (2 * P.k)) - 1u;
    const uint64_t lane_bit = 1ull << lane;

    for (uint32_t w = blockIdx.x * Gm.waves_per_wg + wave; w < Gm.n_windows; w += gridDim.x * Gm.waves_per_wg) {
        const uint64_t off = win_start[w];
        const uint32_t len = win_len[w];
        // aligned 8-byte chunks covering [off, off + len): chunk c holds stream positions [8c, 8c + 8), the
        // window's base j sits at stream position shift + j
        const uint32_t shift = (uint32_t)(off & 7u);
        const uint64_t abase = off - shift;
        const uint32_t n8 = (shift + len + 7u) >> 3;
        __builtin_amdgcn_wave_barrier();       // the previous window's LDS reads are done (same wave: in order)
        uint32_t carry = 0;                    // quality ranks summed over the chunks of earlier rounds
        for (uint32_t c0 = 0; c0 < n8; c0 += kWave) {
            const uint32_t c = c0 + lane;
            uint2 b = make_uint2(0x41414141u, 0x41414141u), q = make_uint2(0x21212121u, 0x21212121u);
            if (c < n8) {
                b = *reinterpret_cast<const uint2 *>(bases + abase + 8u * c);
                q = *reinterpret_cast<const uint2 *>(quals + abase + 8u * c);
            }
            const uint32_t packed = (dna4_pack4(b.x) << 8) | dna4_pack4(b.y);    // 8 bases, the first in the top bits
            // phred94 rank = byte - 33 (utils.h:192-204), byte-wise exact: no borrow runs into the neighbour
            const uint32_t H = 0x80808080u, K = 0x21212121u;
            const uint32_t r0 = ((q.x | H) - K) ^ ((q.x ^ ~K) & H), r1 = ((q.y | H) - K) ^ ((q.y ^ ~K) & H);
            uint32_t before[8], run = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                before[t] = run;
                run += (r0 >> (8 * t)) & 0xFFu;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                before[4 + t] = run;
                run += (r1 >> (8 * t)) & 0xFFu;
            }
            uint32_t incl = run;                // wave scan of the chunk totals
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o, kWave);
                if (lane >= (uint32_t)o) incl += t;
            }
            const uint32_t base_sum = carry + incl - run;
            if (c < n8) {
                pk16[c ^ 1u] = (uint16_t)packed;   // a stream word is big-endian: its first 8 bases are its high half
                *reinterpret_cast<uint4 *>(qsum + 8u * c) =
                    make_uint4(base_sum + before[0], base_sum + before[1], base_sum + before[2], base_sum + before[3]);
                *reinterpret_cast<uint4 *>(qsum + 8u * c + 4u) =
                    make_uint4(base_sum + before[4], base_sum + before[5], base_sum + before[6], base_sum + before[7]);
            }
            carry += __shfl(incl, kWave - 1, kWave);
        }
        if (lane == 0) {
            // the shifts of the last positions read up to two words past the last chunk
            if (n8 & 1u) pk16[n8 ^ 1u] = 0;
            pk[(n8 + 1u) >> 1] = 0;
            pk[((n8 + 1u) >> 1) + 1u] = 0;
            qsum[8u * n8] = carry;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // q-gram starts p = 0 .. len-q; k-mer j owns the q-grams that start at j .. j+G-1 (q_gram_mapper.h:402-403).
        // One bitmap bit per q-gram start, a ballot (scalar) per 64 starts; a round of k-mers needs its own ballot
        // and the next round's first G-1 bits.
        const uint32_t nk = len >= P.k ? len - P.k + 1 : 0;
        const uint32_t nq = nk ? nk + P.G - 1u : 0;
        const uint32_t rounds = (nk + kWave - 1u) / kWave;
        auto ok_ballot = [&](uint32_t rd) -> uint64_t {
            const uint32_t p = rd * kWave + lane;
            const uint32_t at = shift + p, lo = at >> 4, r = at & 15u;
            const uint64_t two = ((uint64_t)pk[lo] << 32) | pk[lo + 1];
            const uint32_t qg = (uint32_t)(two >> (64u - 2u * r - 2u * P.q)) & P.qbits;
            const uint32_t word = BITMAP_LDS ? okmap_lds[qg >> 5] : qgram_ok[qg >> 5];
            return __ballot(p < nq && ((word >> (qg & 31u)) & 1u));
        };
        // k-mers j = 0 .. len-k (views::kmer_hash: size max(len+1,k)-k), 64 per round
        uint32_t n_good = 0;
        uint64_t m0 = rounds ? ok_ballot(0) : 0ull;
        for (uint32_t rd = 0; rd < rounds; rd++) {
            const uint64_t m1 = (rd + 1u) * kWave < nq ? ok_ballot(rd + 1u) : 0ull;
            // bit l of dist: some q-gram starting at j .. j+G-1 is highly distinguishable (wave-uniform, scalar)
            uint64_t dist = m0;
            for (uint32_t g = 1; g < P.G; g++) dist |= (m0 >> g) | (m1 << (64u - g));
            const uint32_t j = rd * kWave + lane;
            const uint32_t at = shift + j, lo = at >> 4, r = at & 15u;
            const uint64_t two = ((uint64_t)pk[lo] << 32) | pk[lo + 1];
            const uint32_t h = (uint32_t)(two >> (64u - 2u * r - 2u * P.k)) & kmask;
            const uint32_t qs = qsum[at + P.k] - qsum[at];                       // quality_filter.h:611-621 (plain sum)
            const bool good = j < nk && (dist & lane_bit) != 0 && qs >= P.minq;  // q_gram_mapper.h:437-438
            const uint64_t m = __ballot(good);
            if (good) goodh[n_good + __popcll(m & (lane_bit - 1ull))] = h;       // ascending j
            n_good += (uint32_t)__popcll(m);
            m0 = m1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // q_gram_mapper.h:445: too few good k-mers -> both candidate lists empty
        if (n_good < P.min_good) {
            if (lane == 0) {
                list_n[w] = 0;
                rows_anded[w] = 0;
            }
            continue;
        }

        // q_gram_mapper.h:457-469: deterministic sample, then the reverse complements of the SAME
        // sampled hashes in the same order; each sample expands to its G contained q-grams' rows.
        uint32_t *list_fwd = row_lists + (size_t)(2 * w) * P.list_len;
        uint32_t *list_rc = list_fwd + P.list_len;
        uint32_t cnt = 0;
        for (uint32_t s = lane; s < P.S; s += kWave) {
            const uint32_t p = pos_table[(size_t)n_good * P.S + s];
            const uint32_t h = goodh[p];
            const uint32_t hr = revcomp_fast(h, P.k);
            for (uint32_t g = 0; g < P.G; g++) {
                // entry g of the sample = its q-gram number row_order[g] (a permutation: the vote ANDs all G of them;
                // the pruning passes that read only the first few want them far apart, bmf_vote2.hip.h)
                const uint32_t gi = (P.row_order >> (4u * g)) & 15u;
                const uint32_t g1 = (h >> (2 * gi)) & P.qbits, g2 = (hr >> (2 * gi)) & P.qbits;
                const int32_t i1 = g1 < P.n_kmers ? k2i[g1] : -1;   // index_of_kmer, q_gram_mapper.h:374-377
                const int32_t i2 = g2 < P.n_kmers ? k2i[g2] : -1;
                list_fwd[s * P.G + g] = i1 >= 0 ? (uint32_t)i1 : P.ones_row;
                list_rc[s * P.G + g] = i2 >= 0 ? (uint32_t)i2 : P.ones_row;
                cnt += (i1 >= 0) + (i2 >= 0);
            }
        }
        // pad both lists with the all-ones row (see bmf_vote_kernel)
        for (uint32_t t = P.S * P.G + lane; t < P.list_len; t += kWave) {
            list_fwd[t] = P.ones_row;
            list_rc[t] = P.ones_row;
        }
        cnt = wave_sum(cnt);
        if (lane == 0) {
            list_n[w] = P.S * P.G;
            rows_anded[w] = cnt;
        }
    }
}

// Gathers the defined candidate ids of the dense per-list slots into one compact array (download path).
__global__ void bmf_compact_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
                                   const uint32_t *__restrict__ buckets, uint32_t max_cand, uint32_t n_items,
                                   uint32_t *__restrict__ compact) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const uint32_t n = counts[item], o = offsets[item];
    for (uint32_t i = 0; i < n; i++) compact[o + i] = buckets[(size_t)item * max_cand + i];
}

// Same for bmf_map_windows' pieces: pack[0] = number of ids, pack[1 + offsets[item] ...] = the item's ids.
__global__ void bmf_compact_total_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
                                         const uint32_t *__restrict__ buckets, uint32_t max_cand, uint32_t n_items,
                                         uint32_t *__restrict__ pack) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const uint32_t n = counts[item], o = offsets[item];
    for (uint32_t i = 0; i < n; i++) pack[1 + o + i] = buckets[(size_t)item * max_cand + i];
    if (item == n_items - 1) pack[0] = o + n;
}

// --------------------------------------------------------------------------------------------------
// vote kernel
// --------------------------------------------------------------------------------------------------

struct u128 {
    uint32_t v[4];
};

__device__ __forceinline__ u128 load_chunk(const uint8_t *p) {
    const uint4 t = *reinterpret_cast<const uint4 *>(p);
    u128 r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}

// fault_tolerate_filter::read (q_gram_mapper.h:75-88) in counter form: every bucket whose AND-ed bit
// is 0 takes one more miss; the PLANES-bit counter saturates at all ones (>= F).  Resets bf to all ones.
template <int CPL, int PLANES>
__device__ __forceinline__ void count_misses(u128 (&bf)[CPL], u128 (&cnt)[PLANES][CPL]) {
#pragma unroll
    for (int j = 0; j < CPL; j++)
#pragma unroll
        for (int x = 0; x < 4; x++) {
            uint32_t sat = cnt[0][j].v[x];
#pragma unroll
            for (int p = 1; p < PLANES; p++) sat &= cnt[p][j].v[x];
            uint32_t carry = ~(bf[j].v[x] | sat);
#pragma unroll
            for (int p = 0; p < PLANES; p++) {
                const uint32_t t = cnt[p][j].v[x] & carry;
                cnt[p][j].v[x] ^= carry;
                carry = t;
            }
            bf[j].v[x] = 0xFFFFFFFFu;
        }
}

// Bits of a lane's 32-bit word that are buckets: only bits < NB exist (std::bitset<NB>), and lanes past
// the end of the row hold nothing.
__device__ __forceinline__ uint32_t bucket_mask(const DevParams &P, uint32_t chunk, int x) {
    const uint32_t b0 = chunk * 128u + (uint32_t)x * 32u;
    if (chunk >= P.n_chunks || b0 >= P.nb) return 0;
    return (P.nb - b0 >= 32u) ? 0xFFFFFFFFu : ((1u << (P.nb - b0)) - 1u);
}

// Bit-sliced "miss count >= F" for one word of PLANES-bit counters (F is wave-uniform).
template <int CPL, int PLANES>
__device__ __forceinline__ uint32_t count_ge(const u128 (&cnt)[PLANES][CPL], int j, int x, uint32_t F) {
    uint32_t ge = 0, eq = 0xFFFFFFFFu;
#pragma unroll
    for (int p = PLANES - 1; p >= 0; p--) {
        const uint32_t c = cnt[p][j].v[x];
        if ((F >> p) & 1u) {
            eq &= c;
        } else {
            ge |= eq & c;
            eq &= ~c;
        }
    }
    return ge | eq;
}

// best_results (q_gram_mapper.h:90-102) + the > max_candidates rule (:471-476): buckets with the
// minimum miss count if that minimum is < F, emitted as ascending ids.  Non-bucket bits carry a
// saturated counter from the start (see the kernel), so they can never be in the minimum set unless
// the minimum itself is saturated (>= F), in which case the result is empty anyway.
//
// SLICED (NB > 65 536): the wave only holds one 65 536-bucket slice of the row.  It then reports its LOCAL
// minimum, and the ids at that minimum (or "more than max_candidates"), to per-(item, slice) slots;
// bmf_merge_slices_kernel takes the minimum over the slices and concatenates the slices that reach it.
//
// cidx[j] = index of the 128-bucket chunk this lane holds in slot j; lanes (and slots) must be in
// ascending chunk order for the emitted ids to come out ascending.
template <int CPL, int PLANES, bool SLICED>
__device__ __forceinline__ void emit_best(const DevParams &P, const u128 (&cnt)[PLANES][CPL], uint32_t item,
                                          uint32_t lane, const uint32_t (&cidx)[CPL], uint32_t *__restrict__ out_counts,
                                          uint32_t *__restrict__ out_buckets, uint32_t *__restrict__ out_min) {
    u128 cand[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++)
#pragma unroll
        for (int x = 0; x < 4; x++) cand[j].v[x] = 0xFFFFFFFFu;
    // most-significant plane first: if some candidate has a 0 in this plane, so does the minimum
    uint32_t m_min = 0;
#pragma unroll
    for (int p = PLANES - 1; p >= 0; p--) {
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < CPL; j++)
#pragma unroll
            for (int x = 0; x < 4; x++) any |= cand[j].v[x] & ~cnt[p][j].v[x];
        if (__ballot(any != 0) != 0) {
#pragma unroll
            for (int j = 0; j < CPL; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) cand[j].v[x] &= ~cnt[p][j].v[x];
        } else {
            m_min |= 1u << p;
        }
    }
    uint32_t pc[CPL], mine = 0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        pc[j] = __popc(cand[j].v[0]) + __popc(cand[j].v[1]) + __popc(cand[j].v[2]) + __popc(cand[j].v[3]);
        mine += pc[j];
    }
    const uint32_t total = wave_sum(mine);
    if (SLICED) {
        if (lane == 0) {
            out_min[item] = m_min;
            out_counts[item] = m_min >= P.F ? 0u : (total > P.max_cand ? P.max_cand + 1u : total);
        }
        if (m_min >= P.F || total > P.max_cand) return;
    } else if (m_min >= P.F || total > P.max_cand) {
        // m_min >= F: every level of the reference's filter is empty.  total > max_cand: cleared.
        if (lane == 0) out_counts[item] = 0;
        return;
    }
    uint32_t *__restrict__ out = out_buckets + (size_t)item * P.max_cand;
    uint32_t base = 0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        if (__ballot(pc[j] != 0) == 0) continue;
        uint32_t incl = pc[j];
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, kWave);
            if (lane >= (uint32_t)o) incl += t;
        }
        uint32_t pos = base + incl - pc[j];
        base += __shfl(incl, kWave - 1, kWave);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            uint32_t bits = cand[j].v[x];
            while (bits) {
                out[pos++] = cidx[j] * 128u + x * 32u + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
            }
        }
    }
    if (!SLICED && lane == 0) out_counts[item] = total;
}

// CPL   : 16-byte chunks per lane (lane l owns chunks l, l+64, ...: every load is 1 KiB contiguous)
// PLANES: bits of the saturating per-bucket miss counter, 2^PLANES-1 >= F
// DEPTH : index rows in flight per wave (register ring)
//
// Branch-free row stream: the row-id list is padded with the all-ones row up to a multiple of DEPTH
// plus DEPTH (a row of ones ANDs as the identity and, landing on a sample boundary, adds no miss), and
// lanes past the end of the row re-read the row's last chunk, so every load is unconditional and the
// compiler can wait for exactly the oldest row in flight (counted vmcnt) instead of draining the ring.
//
// PRUNE (BMF_FLAG_EARLY_EXIT): identical outputs from fewer row bytes.  After F samples, a bucket with
// >= F misses can no longer be in the result (its counter only grows), so (a) when every bucket is dead
// the wave stops, and (b) a lane whose 128-bit chunk holds no live bucket stops loading that chunk --
// typically one lane keeps loading for the true strand, none for the other.  Loads become predicated
// per lane; dead lanes AND stale data into counters that are already >= F, which changes nothing.
template <int CPL, int PLANES, int DEPTH, bool SLICED, bool PRUNE>
__global__ __launch_bounds__(kWave) void bmf_vote_kernel(DevParams P, const uint8_t *__restrict__ rows,
                                                        const uint32_t *__restrict__ row_lists,
                                                        const uint32_t *__restrict__ list_n,
                                                        uint32_t *__restrict__ out_counts,
                                                        uint32_t *__restrict__ out_buckets,
                                                        uint32_t *__restrict__ out_min) {
    const uint32_t lane = threadIdx.x;
    // SLICED: blockIdx.y selects the 65 536-bucket slice; outputs go to the (item, slice) slot
    const uint32_t chunk0 = SLICED ? blockIdx.y * (uint32_t)(CPL * kWave) : 0u;
    const uint32_t item = SLICED ? blockIdx.x * gridDim.y + blockIdx.y : blockIdx.x;   // output slot
    const uint32_t *__restrict__ list = row_lists + (size_t)blockIdx.x * P.list_len;   // blockIdx.x = 2*window + orientation
    if (list_n[blockIdx.x >> 1] == 0) {        // window rejected by the sample kernel
        if (lane == 0) {
            out_counts[item] = 0;
            if (SLICED) out_min[item] = 0xFFFFFFFFu;
        }
        return;
    }
    const uint32_t n_iter = P.list_len - DEPTH;     // rows to consume, a multiple of DEPTH

    uint32_t coff[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        const uint32_t c = chunk0 + lane + kWave * j;
        coff[j] = (c < P.n_chunks ? c : P.n_chunks - 1u) * 16u;
    }

    bool act[CPL];            // PRUNE: this lane still loads chunk j
#pragma unroll
    for (int j = 0; j < CPL; j++) act[j] = chunk0 + lane + kWave * j < P.n_chunks;

    u128 ring[DEPTH][CPL];
    if (PRUNE) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int j = 0; j < CPL; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) ring[d][j].v[x] = 0;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const uint8_t *rp = rows + (size_t)list[d] * P.pitch;
#pragma unroll
        for (int j = 0; j < CPL; j++)
            if (!PRUNE || act[j]) ring[d][j] = load_chunk(rp + coff[j]);
    }

    u128 bf[CPL];             // AND of the current sample's rows (q_gram_mapper.h:400-406)
    u128 cnt[PLANES][CPL];    // bit-sliced miss counters
#pragma unroll
    for (int j = 0; j < CPL; j++)
#pragma unroll
        for (int x = 0; x < 4; x++) {
            bf[j].v[x] = 0xFFFFFFFFu;
            // bits that are not buckets start with a saturated (>= F) counter: never candidates
            const uint32_t dead = ~bucket_mask(P, chunk0 + lane + kWave * j, x);
#pragma unroll
            for (int p = 0; p < PLANES; p++) cnt[p][j].v[x] = dead;
        }

    uint32_t g = 0, samples_done = 0;
    for (uint32_t i = 0; i < n_iter; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int j = 0; j < CPL; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) bf[j].v[x] &= ring[d][j].v[x];
            // refill this ring slot with the row DEPTH positions ahead (list is padded: always valid)
            const uint8_t *rp = rows + (size_t)list[i + d + DEPTH] * P.pitch;
#pragma unroll
            for (int j = 0; j < CPL; j++)
                if (!PRUNE || act[j]) ring[d][j] = load_chunk(rp + coff[j]);
            if (++g == P.G) {
                g = 0;
                count_misses<CPL, PLANES>(bf, cnt);
                ++samples_done;
                // Once every bucket has >= F misses the reference's filter is empty at every level whatever
                // the remaining samples are; chunks without a live bucket need not be read any more.
                if (PRUNE && samples_done >= P.F && samples_done < P.S) {
                    uint32_t alive = 0;
#pragma unroll
                    for (int j2 = 0; j2 < CPL; j2++) {
                        uint32_t a = 0;
#pragma unroll
                        for (int x = 0; x < 4; x++) a |= ~count_ge<CPL, PLANES>(cnt, j2, x, P.F);
                        act[j2] = a != 0;
                        alive |= a;
                    }
                    if (__ballot(alive != 0) == 0) {
                        if (lane == 0) {
                            out_counts[item] = 0;
                            if (SLICED) out_min[item] = 0xFFFFFFFFu;
                        }
                        return;
                    }
                }
            }
        }
    }
    uint32_t cidx[CPL];
#pragma unroll
    for (int j = 0; j < CPL; j++) cidx[j] = chunk0 + lane + kWave * j;
    emit_best<CPL, PLANES, SLICED>(P, cnt, item, lane, cidx, out_counts, out_buckets, out_min);
}

// NB > 65 536: best_results over the slices of one (window, orientation).  One thread per item.
__global__ void bmf_merge_slices_kernel(DevParams P, uint32_t n_items, uint32_t n_slices,
                                        const uint32_t *__restrict__ slice_min, const uint32_t *__restrict__ slice_cnt,
                                        const uint32_t *__restrict__ slice_ids, uint32_t *__restrict__ out_counts,
                                        uint32_t *__restrict__ out_buckets) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    const uint32_t *mn = slice_min + (size_t)item * n_slices, *ct = slice_cnt + (size_t)item * n_slices;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t s = 0; s < n_slices; s++) best = mn[s] < best ? mn[s] : best;
    uint32_t total = 0;
    if (best < P.F)
        for (uint32_t s = 0; s < n_slices; s++)
            if (mn[s] == best) total += ct[s];          // max_cand + 1 marks an overflowing slice
    if (best >= P.F || total > P.max_cand) {
        out_counts[item] = 0;
        return;
    }
    uint32_t *out = out_buckets + (size_t)item * P.max_cand, at = 0;
    for (uint32_t s = 0; s < n_slices; s++)
        if (mn[s] == best) {
            const uint32_t *ids = slice_ids + ((size_t)item * n_slices + s) * P.max_cand;
            for (uint32_t i = 0; i < ct[s]; i++) out[at++] = ids[i];
        }
    out_counts[item] = total;
}

}  // namespace bmf
