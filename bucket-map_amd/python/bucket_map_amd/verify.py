"""ctypes binding of the MI355X alignment verifier (bmv_* in libbmf.so, C ABI in include/bmv.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _bmf_lib

BMV_OK = 0


class BmvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bmv error {code}: {msg}")
        self.code = code


class _Params(C.Structure):
    _fields_ = [("max_query_len", C.c_uint32), ("max_text_len", C.c_uint32), ("device", C.c_int32)]


_u8p, _u32p, _u64p, _i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32))

SYMBOLS = {
    "bmv_last_error": (C.c_char_p, []),
    "bmv_create": (C.c_int, [C.POINTER(_Params), C.POINTER(C.c_void_p)]),
    "bmv_destroy": (None, [C.c_void_p]),
    "bmv_load_genome": (C.c_int, [C.c_void_p, _u8p, C.c_uint64]),
    "bmv_load_genome_records": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32]),
    "bmv_align": (C.c_int, [C.c_void_p, _u8p, C.c_uint64, _u64p, _u32p, _u8p, _u64p, _u32p, C.c_uint32, _u64p]),
    "bmv_results": (C.c_int, [C.c_void_p, _i32p, _u32p, _u64p, _u32p]),
    "bmv_last_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), _u64p]),
}
_ready = False


def lib() -> C.CDLL:
    global _ready
    L = _bmf_lib()
    if not _ready:
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _ready = True
    return L


def _check(rc: int) -> None:
    if rc != BMV_OK:
        raise BmvError(rc, lib().bmv_last_error().decode(errors="replace"))


def _p(a, ty):
    return a.ctypes.data_as(ty)


def cigar_string(packed) -> str:
    return "".join(f"{int(e) >> 4}{'MID'[int(e) & 15]}" for e in packed)


class Verifier:
    """align_pairwise of the BM_ALIGN branch (bucket_locator.h:520-528,569-576) for batches, on one GPU."""

    def __init__(self, max_query_len: int = 65536, max_text_len: int = 81920, device: int = 0):
        h = C.c_void_p()
        prm = _Params(max_query_len, max_text_len, device)
        _check(lib().bmv_create(C.byref(prm), C.byref(h)))
        self._h = h

    def load_genome(self, bases) -> None:
        bases = np.ascontiguousarray(bases, np.uint8)
        _check(lib().bmv_load_genome(self._h, _p(bases, _u8p), len(bases)))

    def align(self, reads, text_start, text_len, text_rc, query_start, query_len):
        """Returns (score i32[n], begin u32[n], cigar_offset u64[n+1], cigar u32[total])."""
        r = np.ascontiguousarray(reads, np.uint8)
        ts, tl = np.ascontiguousarray(text_start, np.uint64), np.ascontiguousarray(text_len, np.uint32)
        trc = np.ascontiguousarray(text_rc, np.uint8)
        qs, ql = np.ascontiguousarray(query_start, np.uint64), np.ascontiguousarray(query_len, np.uint32)
        n = len(ts)
        total = C.c_uint64()
        _check(lib().bmv_align(self._h, _p(r, _u8p), len(r), _p(ts, _u64p), _p(tl, _u32p), _p(trc, _u8p), _p(qs, _u64p),
                               _p(ql, _u32p), n, C.byref(total)))
        score, begin = np.zeros(n, np.int32), np.zeros(n, np.uint32)
        off = np.zeros(n + 1, np.uint64)
        cg = np.zeros(max(total.value, 1), np.uint32)
        _check(lib().bmv_results(self._h, _p(score, _i32p), _p(begin, _u32p), _p(off, _u64p), _p(cg, _u32p)))
        return score, begin, off, cg[: total.value]

    def stats(self) -> dict:
        ms, cells = C.c_float(), C.c_uint64()
        _check(lib().bmv_last_stats(self._h, C.byref(ms), C.byref(cells)))
        return {"ms_kernels": ms.value, "cells": cells.value}

    def close(self) -> None:
        if self._h:
            lib().bmv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
