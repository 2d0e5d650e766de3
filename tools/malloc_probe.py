#!/usr/bin/env python3
"""hipMalloc / hipFree wall time by size in a fresh process (what the verifier's first call pays for its scratch)."""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20)
hip.hipFree(p)
for gb in (1, 4, 16, 45, 45, 90):
    t0 = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30))
    t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    rc2 = hip.hipFree(p)
    t2 = time.perf_counter()
    print(f"{gb:3d} GiB: hipMalloc {1e3 * (t1 - t0):8.1f} ms (rc {rc}), hipFree {1e3 * (t2 - t1):8.1f} ms (rc {rc2})")
