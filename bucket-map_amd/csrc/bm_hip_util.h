// bm_hip_util.h -- small HIP helpers shared by the three translation units of libbmf.so.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

namespace bmhip {

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, device), not to a context: several contexts --
// one per host thread with --gpus, possibly on the same device -- launch the same kernels with different LDS
// sizes.  Only ever raise it, under a lock, so that no context lowers it between another one's call and launch.
inline hipError_t raise_dynamic_lds(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::vector<std::tuple<const void *, int, size_t>> seen;
    int dev = 0;
    hipError_t r = hipGetDevice(&dev);
    if (r != hipSuccess) return r;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : seen)
        if (std::get<0>(e) == fn && std::get<1>(e) == dev) {
            if (bytes <= std::get<2>(e)) return hipSuccess;
            r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (r == hipSuccess) std::get<2>(e) = bytes;
            return r;
        }
    r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (r == hipSuccess) seen.emplace_back(fn, dev, bytes);
    return r;
}

// memcpy by a few threads (a page-cache or heap source into a page-locked buffer runs at 5-8 GB/s per thread)
inline void parallel_memcpy(void *dst, const void *src, size_t bytes, unsigned threads) {
    if (threads <= 1 || bytes < (4u << 20)) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> pool;
    auto part = [&](unsigned t) {
        const size_t lo = bytes * t / threads, hi = bytes * (t + 1) / threads;
        std::memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, hi - lo);
    };
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(part, t);
    part(0);
    for (auto &t : pool) t.join();
}

// A large upload from ORDINARY host memory: hipMemcpy from pageable memory stages through the runtime's own small
// buffers on one thread (4-6 GB/s: 0.3 s for a 1.7 Gbp genome); here the source goes through two page-locked 32 MiB
// buffers, filled by a few threads while the buffer before is on the link.  Synchronous: returns when the data is there.
inline hipError_t upload_pageable(void *dst, const void *src, size_t bytes) {
    constexpr size_t kPiece = 32u << 20;
    if (bytes < 2 * kPiece) return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc(&stage[i], kPiece, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    const unsigned threads = std::max(1u, std::min(3u, std::thread::hardware_concurrency() / 2u));
    size_t piece = 0;
    for (size_t at = 0; e == hipSuccess && at < bytes; at += kPiece, piece++) {
        const int slot = (int)(piece & 1);
        const size_t n = std::min(kPiece, bytes - at);
        if (piece >= 2) e = hipEventSynchronize(done[slot]);           // the buffer's previous copy has left it
        if (e != hipSuccess) break;
        parallel_memcpy(stage[slot], static_cast<const char *>(src) + at, n, threads);
        e = hipMemcpyAsync(static_cast<char *>(dst) + at, stage[slot], n, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(done[slot], stream);
    }
    if (stream) {
        const hipError_t s = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = s;
    }
    for (int i = 0; i < 2; i++) {
        if (done[i]) (void)hipEventDestroy(done[i]);
        if (stage[i]) (void)hipHostFree(stage[i]);
    }
    if (stream) (void)hipStreamDestroy(stream);
    return e;
}

// The same for a source that lies in PIECES on the host (the records of a genome, each a buffer of its own): the pieces go
// to dst back to back, in order, through the same two page-locked buffers -- no flattened copy on the host.
inline hipError_t upload_pageable_records(void *dst, const uint8_t *const *rec, const uint64_t *rec_len, uint32_t n_records) {
    constexpr size_t kPiece = 32u << 20;
    uint64_t total = 0;
    for (uint32_t r = 0; r < n_records; r++) total += rec_len[r];
    if (total == 0) return hipSuccess;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc(&stage[i], kPiece, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    const unsigned threads = std::max(1u, std::min(3u, std::thread::hardware_concurrency() / 2u));
    uint32_t r = 0;                 // the record and the offset in it where the next piece begins
    uint64_t in_r = 0;
    size_t piece = 0;
    for (uint64_t at = 0; e == hipSuccess && at < total; at += kPiece, piece++) {
        const int slot = (int)(piece & 1);
        const size_t n = (size_t)std::min<uint64_t>(kPiece, total - at);
        if (piece >= 2) e = hipEventSynchronize(done[slot]);           // the buffer's previous copy has left it
        if (e != hipSuccess) break;
        // the stretches of records that make up this piece
        struct Part { const uint8_t *src; size_t to, len; };
        std::vector<Part> parts;
        for (size_t filled = 0; filled < n;) {
            while (in_r == rec_len[r]) {
                r++;
                in_r = 0;
            }
            const size_t take = (size_t)std::min<uint64_t>(rec_len[r] - in_r, n - filled);
            parts.push_back(Part{rec[r] + in_r, filled, take});
            filled += take;
            in_r += take;
        }
        // ... copied by a few threads, each a contiguous share of the piece's bytes
        auto share = [&](unsigned t) {
            const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
            for (const Part &p : parts) {
                const size_t a = std::max(lo, p.to), b = std::min(hi, p.to + p.len);
                if (a < b) std::memcpy(static_cast<char *>(stage[slot]) + a, p.src + (a - p.to), b - a);
            }
        };
        if (threads <= 1 || n < (4u << 20)) {
            for (unsigned t = 0; t < threads; t++) share(t);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < threads; t++) pool.emplace_back(share, t);
            share(0);
            for (auto &t : pool) t.join();
        }
        e = hipMemcpyAsync(static_cast<char *>(dst) + at, stage[slot], n, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(done[slot], stream);
    }
    if (stream) {
        const hipError_t s = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = s;
    }
    for (int i = 0; i < 2; i++) {
        if (done[i]) (void)hipEventDestroy(done[i]);
        if (stage[i]) (void)hipHostFree(stage[i]);
    }
    if (stream) (void)hipStreamDestroy(stream);
    return e;
}

}  // namespace bmhip
