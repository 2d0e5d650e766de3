#!/usr/bin/env python3
"""How long the library's start-up steps take in a fresh process: dlopen (code objects registered), the first HIP
call, a filter context.  `python tools/init_probe.py`"""
import ctypes
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
t0 = time.perf_counter()
lib = ctypes.CDLL(os.path.join(ROOT, "bucket-map_amd", "libbmf.so"))
t1 = time.perf_counter()
hip = ctypes.CDLL("libamdhip64.so")
n = ctypes.c_int(0)
hip.hipGetDeviceCount(ctypes.byref(n))
t2 = time.perf_counter()
hip.hipSetDevice(0)
hip.hipFree(None)
t3 = time.perf_counter()
sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))
import numpy as np
import bucket_map_amd as bma
t4 = time.perf_counter()
f = bma.Filter(bma.Params.from_cli(71, read_len=150))
t5 = time.perf_counter()
print(f"dlopen libbmf.so {t1 - t0:.3f} s, hipGetDeviceCount {t2 - t1:.3f} s, hipSetDevice+hipFree(0) {t3 - t2:.3f} s, "
      f"import binding {t4 - t3:.3f} s, first context {t5 - t4:.3f} s")
