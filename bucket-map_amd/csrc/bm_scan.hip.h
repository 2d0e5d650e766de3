// bm_scan.hip.h -- device-wide exclusive prefix sum, hand-written for gfx950 (64-lane waves, DPP row moves through
// __shfl_up): the one "library primitive" the filter's packed output, the verifier's CIGAR gather and the locator's
// candidate grouping need.  Three short launches on the caller's stream:
//   block_sums : each 256-thread block adds up its 2 048 items
//   scan_sums  : one block scans the block sums in place (exclusive)
//   scan_items : each block scans its items again and adds its block's offset; writes n + 1 values -- out[n] is the total
// Counts are 32-bit, offsets 32- or 64-bit (TOut).  `tmp` holds ceil(n / 2048) + 1 values of TOut.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmscan {

constexpr int kThreads = 256;
constexpr int kItems = 8;                       // per thread
constexpr int kTile = kThreads * kItems;        // per block

inline size_t tmp_elems(uint64_t n) { return (size_t)((n + kTile - 1) / kTile) + 1; }

template <typename T>
__device__ __forceinline__ T wave_inclusive(T v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    return v;
}

// inclusive scan of one value per thread over the block; *total receives the block's sum
template <typename T>
__device__ __forceinline__ T block_inclusive(T v, T *wave_sums /* kThreads / 64 + 1 */, T *total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    T incl = wave_inclusive(v, lane);
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    T before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) {
        const T s = wave_sums[w];
        if ((uint32_t)w < wave) before += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return incl + before;
}

template <typename TOut>
__global__ __launch_bounds__(kThreads) void block_sums_kernel(const uint32_t *__restrict__ in, uint64_t n, TOut *__restrict__ sums) {
    __shared__ TOut ws[kThreads / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
    TOut v = 0;
#pragma unroll
    for (int i = 0; i < kItems; i++)
        if (base + i < n) v += in[base + i];
    TOut total;
    (void)block_inclusive<TOut>(v, ws, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

template <typename TOut>
__global__ __launch_bounds__(kThreads) void scan_sums_kernel(TOut *__restrict__ sums, uint64_t n_blocks) {
    __shared__ TOut ws[kThreads / 64 + 1];
    TOut carry = 0;
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += kThreads) {
        const uint64_t i = b0 + threadIdx.x;
        const TOut v = i < n_blocks ? sums[i] : (TOut)0;
        TOut total;
        const TOut incl = block_inclusive<TOut>(v, ws, &total);
        if (i < n_blocks) sums[i] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) sums[n_blocks] = carry;
}

template <typename TOut>
__global__ __launch_bounds__(kThreads) void scan_items_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                            const TOut *__restrict__ sums, TOut *__restrict__ out) {
    __shared__ TOut ws[kThreads / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
    uint32_t item[kItems];
    TOut v = 0;
#pragma unroll
    for (int i = 0; i < kItems; i++) {
        item[i] = base + i < n ? in[base + i] : 0u;
        v += item[i];
    }
    TOut total;
    TOut run = block_inclusive<TOut>(v, ws, &total) - v + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kItems; i++) {
        if (base + i < n) out[base + i] = run;
        run += item[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = sums[gridDim.x];
}

// out[i] = in[0] + ... + in[i-1] for i in 0..n (n + 1 values).  n == 0 writes out[0] = 0.
template <typename TOut>
inline hipError_t exclusive_sum(const uint32_t *in, TOut *out, uint64_t n, TOut *tmp, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(out, 0, sizeof(TOut), stream);
    const uint64_t n_blocks = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(block_sums_kernel<TOut>, dim3((unsigned)n_blocks), dim3(kThreads), 0, stream, in, n, tmp);
    hipLaunchKernelGGL(scan_sums_kernel<TOut>, dim3(1), dim3(kThreads), 0, stream, tmp, n_blocks);
    hipLaunchKernelGGL(scan_items_kernel<TOut>, dim3((unsigned)n_blocks), dim3(kThreads), 0, stream, in, n, tmp, out);
    return hipGetLastError();
}

}  // namespace bmscan
