// device_pool.h -- the one way the host side spreads a call over several GPUs: contiguous ranges of independent
// units (windows, candidates, alignments), one host thread per device, results written in place so that their
// order is the single-device order.  No collective anywhere: the units never talk to each other.
#pragma once

#include <cstddef>
#include <cstdint>
#include <exception>
#include <thread>
#include <vector>

namespace bm {

// fn(d) for d = 0..D-1, each on its own thread (inline when D == 1); the first exception is rethrown here.
template <class F>
inline void for_each_device(size_t D, F &&fn) {
    if (D == 1) {
        fn(size_t{0});
        return;
    }
    std::vector<std::exception_ptr> err(D);
    std::vector<std::thread> pool;
    pool.reserve(D);
    for (size_t d = 0; d < D; d++)
        pool.emplace_back([&, d]() {
            try {
                fn(d);
            } catch (...) {
                err[d] = std::current_exception();
            }
        });
    for (auto &t : pool) t.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

// cut[d] .. cut[d + 1] = the units of device d: equal counts
inline std::vector<uint32_t> cut_evenly(uint32_t n, size_t D) {
    std::vector<uint32_t> cut(D + 1);
    for (size_t d = 0; d <= D; d++) cut[d] = static_cast<uint32_t>(static_cast<uint64_t>(n) * d / D);
    return cut;
}

// ... equal summed cost (cost(i) >= 0), for units of very different size
template <class Cost>
inline std::vector<uint32_t> cut_by_cost(uint32_t n, size_t D, Cost &&cost) {
    std::vector<uint32_t> cut(D + 1, n);
    cut[0] = 0;
    if (D == 1) return cut;
    long double total = 0;
    for (uint32_t i = 0; i < n; i++) total += static_cast<long double>(cost(i));
    long double acc = 0;
    size_t d = 1;
    for (uint32_t i = 0; i < n && d < D; i++) {
        acc += static_cast<long double>(cost(i));
        while (d < D && acc >= total * d / D) cut[d++] = i + 1;
    }
    return cut;
}

}  // namespace bm
