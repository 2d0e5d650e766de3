// bm_hip_util.h -- small HIP helpers shared by the three translation units of libbmf.so.
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
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

}  // namespace bmhip
