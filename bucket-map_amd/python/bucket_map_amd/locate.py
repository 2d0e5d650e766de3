"""ctypes binding of the MI355X locator candidate scan (bml_* in libbmf.so, C ABI in include/bml.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _bmf_lib

BML_OK = 0


class BmlError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bml error {code}: {msg}")
        self.code = code


class _Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("num_samples", C.c_uint32), ("allowed_mismatch", C.c_int32),
                ("allowed_indel", C.c_int32), ("max_bucket_bases", C.c_uint32), ("device", C.c_int32)]


_u8p, _u16p, _u32p, _u64p, _i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int32))

SYMBOLS = {
    "bml_last_error": (C.c_char_p, []),
    "bml_create": (C.c_int, [C.POINTER(_Params), C.POINTER(C.c_void_p)]),
    "bml_destroy": (None, [C.c_void_p]),
    "bml_load_genome": (C.c_int, [C.c_void_p, _u8p, C.c_uint64, _u64p, _u32p, C.c_uint32]),
    "bml_load_genome_records": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _u64p, C.c_uint32, _u64p, _u32p, C.c_uint32]),
    "bml_sample_windows": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_uint64, _u64p, _u32p, C.c_uint32, C.c_uint32, _u32p, _u16p, _u8p]),
    "bml_sample_text_windows": (C.c_int, [C.c_void_p, _u8p, C.c_uint64, _u64p, _u64p, _u32p, C.c_uint32, C.c_uint32, _u32p, _u16p, _u8p]),
    "bml_locate": (C.c_int, [C.c_void_p, _u32p, _u16p, _u32p, C.c_uint32, _u32p, _u32p, _u8p, C.c_uint32, _i32p, _u32p]),
    "bml_last_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), _u64p]),
    "bml_last_heavy_candidates": (C.c_int, [C.c_void_p, _u32p]),
    "bml_last_count_histogram": (C.c_int, [C.c_void_p, C.c_uint32, _u64p, _u64p]),
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
    if rc != BML_OK:
        raise BmlError(rc, lib().bml_last_error().decode(errors="replace"))


def _p(a, ty):
    return a.ctypes.data_as(ty)


class LocatorScan:
    """_create_kmer_index + _find_offset for batches of candidates, on one GPU (bml_ctx)."""

    def __init__(self, k: int, num_samples: int, allowed_mismatch: int, allowed_indel: int, max_bucket_bases: int,
                 device: int = 0):
        self.k, self.p = k, num_samples
        h = C.c_void_p()
        prm = _Params(k, num_samples, allowed_mismatch, allowed_indel, max_bucket_bases, device)
        _check(lib().bml_create(C.byref(prm), C.byref(h)))
        self._h = h

    def load_genome(self, bases, bucket_start, bucket_len) -> None:
        bases = np.ascontiguousarray(bases, np.uint8)
        bs = np.ascontiguousarray(bucket_start, np.uint64)
        bl = np.ascontiguousarray(bucket_len, np.uint32)
        _check(lib().bml_load_genome(self._h, _p(bases, _u8p), len(bases), _p(bs, _u64p), _p(bl, _u32p), len(bs)))

    def load_genome_records(self, records, bucket_start, bucket_len) -> None:
        """bml_load_genome_records: the genome as a list of uint8 arrays (records), uploaded back to back."""
        recs = [np.ascontiguousarray(r, np.uint8) for r in records]
        ptrs = (C.c_void_p * len(recs))(*[r.ctypes.data for r in recs])
        lens = np.array([len(r) for r in recs], np.uint64)
        bs = np.ascontiguousarray(bucket_start, np.uint64)
        bl = np.ascontiguousarray(bucket_len, np.uint32)
        _check(lib().bml_load_genome_records(self._h, ptrs, _p(lens, _u64p), len(recs), _p(bs, _u64p), _p(bl, _u32p), len(bs)))

    def sample_windows(self, bases, quals, win_start, win_len, min_base_quality: int):
        """_prepare_read_query's sampling: (hash u32[n, p], pos u16[n, p], has u8[n])."""
        b, q = np.ascontiguousarray(bases, np.uint8), np.ascontiguousarray(quals, np.uint8)
        ws, wl = np.ascontiguousarray(win_start, np.uint64), np.ascontiguousarray(win_len, np.uint32)
        n = len(ws)
        h = np.zeros((n, self.p), np.uint32)
        pos = np.zeros((n, self.p), np.uint16)
        has = np.zeros(n, np.uint8)
        _check(lib().bml_sample_windows(self._h, _p(b, _u8p), _p(q, _u8p), len(b), _p(ws, _u64p), _p(wl, _u32p), n,
                                        min_base_quality, _p(h, _u32p), _p(pos, _u16p), _p(has, _u8p)))
        return h, pos, has

    def sample_text_windows(self, text, seq_start, qual_start, win_len, min_base_quality: int):
        """bml_sample_text_windows: the same sampling for windows whose bases and qualities lie apart in one buffer."""
        t = np.ascontiguousarray(text, np.uint8)
        ss, qs = np.ascontiguousarray(seq_start, np.uint64), np.ascontiguousarray(qual_start, np.uint64)
        wl = np.ascontiguousarray(win_len, np.uint32)
        n = len(ss)
        h = np.zeros((n, self.p), np.uint32)
        pos = np.zeros((n, self.p), np.uint16)
        has = np.zeros(n, np.uint8)
        _check(lib().bml_sample_text_windows(self._h, _p(t, _u8p), len(t), _p(ss, _u64p), _p(qs, _u64p), _p(wl, _u32p), n,
                                             min_base_quality, _p(h, _u32p), _p(pos, _u16p), _p(has, _u8p)))
        return h, pos, has

    def locate(self, sample_hash, sample_pos, seg_len, pair_bucket, pair_window, pair_rc):
        sh = np.ascontiguousarray(sample_hash, np.uint32)
        sp = np.ascontiguousarray(sample_pos, np.uint16)
        sl = np.ascontiguousarray(seg_len, np.uint32)
        pb = np.ascontiguousarray(pair_bucket, np.uint32)
        pw = np.ascontiguousarray(pair_window, np.uint32)
        pr = np.ascontiguousarray(pair_rc, np.uint8)
        n = len(pb)
        off = np.full(n, -1, np.int32)
        votes = np.zeros(n, np.uint32)
        _check(lib().bml_locate(self._h, _p(sh, _u32p), _p(sp, _u16p), _p(sl, _u32p), len(sl), _p(pb, _u32p),
                                _p(pw, _u32p), _p(pr, _u8p), n, _p(off, _i32p), _p(votes, _u32p)))
        return off, votes

    def stats(self) -> dict:
        a, b, c, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint64()
        _check(lib().bml_last_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        h = C.c_uint32()
        _check(lib().bml_last_heavy_candidates(self._h, C.byref(h)))
        return {"ms_scan": a.value, "ms_host": b.value, "ms_replay": c.value, "occurrences": n.value, "heavy_candidates": h.value}

    def count_histogram(self, n_pairs: int):
        """(candidates, occurrences) of the last locate by occurrence count: bin b = [2^(b-1), 2^b), bin 0 = none."""
        a, b = np.zeros(33, np.uint64), np.zeros(33, np.uint64)
        _check(lib().bml_last_count_histogram(self._h, n_pairs, _p(a, _u64p), _p(b, _u64p)))
        return a, b

    def close(self) -> None:
        if self._h:
            lib().bml_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
