// bmi_kernels.hip.h -- gfx950 kernels that build the q-gram x bucket index directly in HBM.
//
// GPU form of bucket_indexer::index / _insert_into_bucket (bucket_map/indexer/bucket_indexer.h:49-61,
// :170-216): bit (row of q-gram g, bucket b) is set iff q-gram g occurs in bucket b.  Two passes:
//   bmi_presence_kernel  : one workgroup per bucket; the bucket's q-gram PRESENCE bitmap (4^q bits, 32 KiB
//                          for q = 9) is accumulated in LDS with LDS atomics and written out once,
//                          coalesced -- a (bucket x q-gram) bit matrix, the transpose of the index.
//   bmi_transpose_kernel : 64 x 64 bit tiles are transposed with 64 wave ballots each and stored into the
//                          index rows (q-gram x bucket), skipping q-grams FracMinHash did not keep.
// The reference does the same work as one std::bitset::set per base of every bucket on one CPU thread.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bm_dna4.hip.h"

namespace bmi {

constexpr int kThreads = 1024;

// LDS (dynamic): bitmap[4^q / 32] u32 | packed[stream_words] u32 (a segment of the bucket as a 2-bit stream, 16 bases
// per word, the first base in the top bits: aligned 16-byte loads and register folding instead of a byte load and a table
// look-up per base; a segment may start at any byte, its base j sits at stream position shift + j).  A bucket that does
// not fit beside the bitmap goes through in segments of seg_bases bases that overlap by q - 1.
__global__ __launch_bounds__(kThreads) void bmi_presence_kernel(const uint8_t *__restrict__ genome,
                                                               const uint64_t *__restrict__ bucket_start,
                                                               const uint32_t *__restrict__ bucket_len,
                                                               const uint8_t *__restrict__ dna4_lut, uint32_t q,
                                                               uint32_t stream_words, uint32_t seg_bases,
                                                               uint32_t *__restrict__ presence /* n_buckets x 4^q/32 */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t n_words = (1u << (2 * q)) >> 5;
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(smem);
    uint32_t *packed = bitmap + n_words;
    (void)dna4_lut;
    const uint32_t tid = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const uint32_t len = bucket_len[b];
    const uint32_t mask = (uint32_t)((1ull << (2 * q)) - 1ull);
    for (uint32_t w = tid; w < n_words; w += kThreads) bitmap[w] = 0;
    for (uint32_t s0 = 0; s0 + q <= len; s0 += seg_bases - (q - 1u)) {
        const uint32_t seg = len - s0 < seg_bases ? len - s0 : seg_bases;       // bases of this segment
        const uint64_t start = bucket_start[b] + s0;
        const uint32_t shift = (uint32_t)(start & 15u);
        const uint8_t *abase = genome + (start - shift);
        const uint32_t nw = (shift + seg + 15u) / 16u;
        __syncthreads();                                                         // the previous segment has been read
        for (uint32_t w = tid; w < stream_words; w += kThreads)
            packed[w] = w < nw ? bmdna::dna4_pack16(*reinterpret_cast<const uint4 *>(abase + 16u * w)) : 0u;
        __syncthreads();
        const uint32_t nq = seg - q + 1;
        for (uint32_t j = tid; j < nq; j += kThreads) {
            const uint32_t at = shift + j;
            const uint64_t two = ((uint64_t)packed[at >> 4] << 32) | packed[(at >> 4) + 1];
            const uint32_t h = (uint32_t)(two >> (64u - 2u * (at & 15u) - 2u * q)) & mask;
            atomicOr(&bitmap[h >> 5], 1u << (h & 31u));
        }
        if (seg < seg_bases) break;                                              // that was the bucket's last base
    }
    __syncthreads();
    uint32_t *out = presence + (size_t)b * n_words;
    for (uint32_t w = tid; w < n_words; w += kThreads) out[w] = bitmap[w];
}

// One wave per (group of 64 buckets, 64 consecutive q-grams).  presence is read as u64 words.
// Grid: x = 16-wave workgroups over the q-gram words of one group, y = groups of 64 buckets (strided: a HIP grid
// dimension times its block dimension must stay below 2^32 threads, and y below 65 536 workgroups).
__global__ __launch_bounds__(kThreads) void bmi_transpose_kernel(const uint64_t *__restrict__ presence, uint32_t n_buckets,
                                                                uint32_t q, const int32_t *__restrict__ k2i,
                                                                uint8_t *__restrict__ rows, uint32_t pitch) {
    const uint32_t n_w64 = (1u << (2 * q)) >> 6;           // 64-bit words per bucket
    const uint32_t w64 = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_groups = (n_buckets + 63u) / 64u;
    if (w64 >= n_w64) return;
    const int32_t row = k2i[w64 * 64u + lane];             // FracMinHash: -1 = q-gram not kept
    for (uint32_t group = blockIdx.y; group < n_groups; group += gridDim.y) {
        const uint32_t b = group * 64u + lane;
        const uint64_t v = b < n_buckets ? presence[(size_t)b * n_w64 + w64] : 0ull;
        uint64_t mine = 0;
#pragma unroll 8
        for (uint32_t t = 0; t < 64; t++) {
            const uint64_t m = __ballot((v >> t) & 1ull);  // bit i = bucket group*64+i holds q-gram w64*64+t
            if (lane == t) mine = m;
        }
        if (row >= 0) *reinterpret_cast<uint64_t *>(rows + (size_t)row * pitch + (size_t)group * 8u) = mine;
    }
}

}  // namespace bmi
