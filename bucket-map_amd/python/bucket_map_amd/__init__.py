"""ctypes binding of the MI355X candidate-bucket filter (libbmf.so, C ABI in include/bmf.h).

This is the reference-side stub a Python caller would use; it holds no arithmetic of its own.  The
library is looked up IN-TREE (bucket-map_amd/libbmf.so) and loading fails loudly when it has not been
built: there is no CPU fallback for the filter.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_PKG_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REPO_ROOT = os.path.abspath(os.path.join(_PKG_ROOT, ".."))
LIBBMF_PATH = os.path.join(_PKG_ROOT, "libbmf.so")

BMF_FLAG_EARLY_EXIT = 1
BMF_OK, BMF_ERR_ARG, BMF_ERR_HIP, BMF_ERR_STATE, BMF_ERR_IO, BMF_ERR_UNSUPPORTED = range(6)


class BmfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bmf error {code}: {msg}")
        self.code = code


class _Params(C.Structure):
    _fields_ = [
        ("num_buckets", C.c_uint32), ("q", C.c_uint32), ("k", C.c_uint32), ("num_samples", C.c_uint32),
        ("num_fault", C.c_uint32), ("threshold", C.c_uint32), ("min_base_quality", C.c_uint32),
        ("max_candidates", C.c_uint32), ("read_len", C.c_uint32), ("num_segment_samples", C.c_uint32),
        ("device", C.c_int32), ("flags", C.c_uint32),
    ]


_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)

# every symbol include/bmf.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "bmf_abi_version": (C.c_int, []),
    "bmf_last_error": (C.c_char_p, []),
    "bmf_fault_from_rate": (C.c_uint32, [C.c_uint32, C.c_float]),
    "bmf_threshold": (C.c_uint32, [C.c_float, C.c_uint32]),
    "bmf_ceil_mul_f32": (C.c_uint32, [C.c_float, C.c_uint32]),
    "bmf_create": (C.c_int, [C.POINTER(_Params), C.POINTER(C.c_void_p)]),
    "bmf_destroy": (None, [C.c_void_p]),
    "bmf_load_index": (C.c_int, [C.c_void_p, _u8p, C.c_uint64, _i32p, C.c_uint64]),
    "bmf_build_index": (C.c_int, [C.c_void_p, _u8p, C.c_uint64, _u64p, _u32p, C.c_uint32, _i32p, C.c_uint64]),
    "bmf_index_download": (C.c_int, [C.c_void_p, _u8p, _u64p]),
    "bmf_load_index_files": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "bmf_reset": (C.c_int, [C.c_void_p]),
    "bmf_index_zeros": (C.c_int, [C.c_void_p, _u32p]),
    "bmf_window_starts": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, _u32p]),
    "bmf_map_windows": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_uint64, _u64p, _u32p, C.c_uint32, _u32p, _u32p]),
    "bmf_map_windows_compact": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_uint64, _u64p, _u32p, C.c_uint32, _u32p, _u32p,
                                          C.c_uint64, _u64p]),
    "bmf_map_reserve": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int]),
    "bmf_map_text_windows_compact": (C.c_int, [C.c_void_p, _u8p, C.c_uint64, _u64p, _u64p, _u32p, C.c_uint32, _u32p, _u32p,
                                               C.c_uint64, _u64p]),
    "bmf_batch_create": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_uint64, _u64p, _u32p, C.c_uint32,
                                   C.POINTER(C.c_void_p)]),
    "bmf_batch_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bmf_batch_download": (C.c_int, [C.c_void_p, C.c_void_p, _u32p, _u32p]),
    "bmf_batch_rows_anded": (C.c_int, [C.c_void_p, C.c_void_p, _u64p]),
    "bmf_batch_destroy": (None, [C.c_void_p, C.c_void_p]),
    "bmf_sync": (C.c_int, [C.c_void_p]),
    "bmf_device_memory": (C.c_int, [C.c_int, _u64p, _u64p]),
    "bmf_profile_begin": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bmf_profile_end": (C.c_int, [C.c_void_p, _u32p, _f32p, _f32p]),
    "bmf_pinned_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "bmf_pinned_free": (None, [C.c_void_p]),
    "bmf_info": (C.c_int, [C.c_void_p, _u32p, _u32p, _u32p, _u32p]),
    "bmf_pass1_rows": (C.c_int, [C.c_void_p, _u32p]),
    "bmf_pass1_fold": (C.c_int, [C.c_void_p, _u32p, _u32p]),
    "bmf_batch_pass2_counts": (C.c_int, [C.c_void_p, C.c_void_p, _u32p, _u32p]),
    "bmf_batch_recount_loads": (C.c_int, [C.c_void_p, C.c_void_p, _u64p]),
    "bmf_batch_live_histogram": (C.c_int, [C.c_void_p, C.c_void_p, _u64p, _u64p]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads libbmf.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBBMF_PATH):
            raise ImportError(
                f"{LIBBMF_PATH} is missing: build it with `make` or `python -c 'import __graft_entry__ as g; "
                "g.build()'`.  The filter has no CPU fallback.")
        L = C.CDLL(LIBBMF_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != BMF_OK:
        raise BmfError(rc, lib().bmf_last_error().decode(errors="replace"))


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


@dataclass
class Params:
    """Run-time form of q_gram_mapper's constructor arguments (q_gram_mapper.h:281-286)."""
    num_buckets: int
    q: int = 9
    k: int = 12
    num_samples: int = 15
    num_fault: int = 6
    threshold: int = 0
    min_base_quality: int = 300
    max_candidates: int = 30
    read_len: int = 300
    num_segment_samples: int = 5
    device: int = 0
    flags: int = 0

    @classmethod
    def from_cli(cls, num_buckets: int, *, index_seed: int = 9, query_seed: int = 12, read_len: int = 300,
                 mapper_samples: int = 15, max_error_rate: float = 0.4, distinguishability: float = 0.5,
                 average_base_quality: int = 25, device: int = 0, flags: int = 0) -> "Params":
        """Derives F, threshold and min_base_quality exactly as main.cpp:202-209 does (float32)."""
        L = lib()
        return cls(num_buckets=num_buckets, q=index_seed, k=query_seed, num_samples=mapper_samples,
                   num_fault=L.bmf_fault_from_rate(mapper_samples, max_error_rate),
                   threshold=L.bmf_threshold(distinguishability, num_buckets),
                   min_base_quality=average_base_quality * query_seed, read_len=read_len, device=device, flags=flags)

    def to_c(self) -> _Params:
        return _Params(self.num_buckets, self.q, self.k, self.num_samples, self.num_fault, self.threshold,
                       self.min_base_quality, self.max_candidates, self.read_len, self.num_segment_samples,
                       self.device, self.flags)


def windows_for_reads(offsets, read_len: int, n_seg: int = 5):
    """q_gram_mapper::map's windowing (q_gram_mapper.h:510-523) for reads stored back to back:
    read r = [offsets[r], offsets[r+1]).  Returns (win_start u64, win_len u32, read_id u32,
    start_in_read u32): one window [0, min(read_len, len)) per read, or n_seg windows at
    Sampler(n_seg) positions for reads longer than 2*read_len."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    lens = np.diff(offsets).astype(np.int64)
    n = len(lens)
    long_reads = np.nonzero(lens > 2 * read_len)[0]
    if long_reads.size == 0:
        return (offsets[:-1].copy(), np.minimum(lens, read_len).astype(np.uint32), np.arange(n, dtype=np.uint32),
                np.zeros(n, dtype=np.uint32))
    ws, wl, rid, sir = [], [], [], []
    buf = (C.c_uint32 * max(n_seg, 1))()
    is_long = np.zeros(n, bool)
    is_long[long_reads] = True
    for r in range(n):
        if is_long[r]:
            m = lib().bmf_window_starts(int(lens[r]), read_len, n_seg, buf)
            starts = [buf[i] for i in range(m)]
        else:
            starts = [0]
        for st in starts:
            ws.append(int(offsets[r]) + st)
            wl.append(min(st + read_len, int(lens[r])) - st)
            rid.append(r)
            sir.append(st)
    return (np.array(ws, np.uint64), np.array(wl, np.uint32), np.array(rid, np.uint32), np.array(sir, np.uint32))


class PinnedArray:
    """A uint8 numpy array in page-locked host memory (bmf_pinned_alloc): `.array` is valid until close()."""

    def __init__(self, n: int):
        self._p = C.c_void_p()
        _check(lib().bmf_pinned_alloc(max(1, n), C.byref(self._p)))
        self.array = np.ctypeslib.as_array(C.cast(self._p, _u8p), shape=(n,))

    def close(self) -> None:
        if self._p:
            self.array = None
            lib().bmf_pinned_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_copy(a) -> PinnedArray:
    """Copy of a byte array in page-locked memory: the source bmf_map_windows can overlap with its kernels."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = PinnedArray(a.size)
    out.array[:] = a
    return out


class Batch:
    """Device-resident batch of windows (bmf_batch)."""

    def __init__(self, flt: "Filter", bases, quals, win_start, win_len):
        self._flt = flt
        self.n_windows = len(win_start)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        win_start = np.ascontiguousarray(win_start, dtype=np.uint64)
        win_len = np.ascontiguousarray(win_len, dtype=np.uint32)
        assert len(win_len) == self.n_windows and len(quals) == len(bases)
        h = C.c_void_p()
        _check(lib().bmf_batch_create(flt._h, _ptr(bases, _u8p), _ptr(quals, _u8p), len(bases),
                                      _ptr(win_start, _u64p), _ptr(win_len, _u32p), self.n_windows, C.byref(h)))
        self._h = h

    def run(self) -> None:
        _check(lib().bmf_batch_run(self._flt._h, self._h))

    def download(self):
        mc = self._flt.params.max_candidates
        counts = np.zeros((self.n_windows, 2), dtype=np.uint32)
        buckets = np.zeros((self.n_windows, 2, mc), dtype=np.uint32)
        _check(lib().bmf_batch_download(self._flt._h, self._h, _ptr(counts, _u32p), _ptr(buckets, _u32p)))
        return counts, buckets

    def rows_anded(self) -> int:
        v = C.c_uint64()
        _check(lib().bmf_batch_rows_anded(self._flt._h, self._h, C.byref(v)))
        return int(v.value)

    def pass2_counts(self):
        """(items recounted by the packed kernel, items on the slow path) of the last run; two-pass pruning only."""
        a, b = C.c_uint32(), C.c_uint32()
        _check(lib().bmf_batch_pass2_counts(self._flt._h, self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def live_histogram(self):
        """(stored, lowest): items by stored chunks after pass 1 / by chunks at their lowest level; [33] = slow items."""
        a, b = np.zeros(34, np.uint64), np.zeros(34, np.uint64)
        _check(lib().bmf_batch_live_histogram(self._flt._h, self._h, _ptr(a, _u64p), _ptr(b, _u64p)))
        return a, b

    def recount_loads(self) -> int:
        """16-byte column loads (one 64-byte sector each) of the recount kernel in the last run; two-pass pruning only."""
        v = C.c_uint64()
        _check(lib().bmf_batch_recount_loads(self._flt._h, self._h, C.byref(v)))
        return int(v.value)

    def close(self) -> None:
        if self._h:
            lib().bmf_batch_destroy(self._flt._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Filter:
    """The candidate-bucket filter on one GPU (bmf_ctx): load / map_windows / reset."""

    def __init__(self, params: Params):
        self.params = params
        h = C.c_void_p()
        cp = params.to_c()
        _check(lib().bmf_create(C.byref(cp), C.byref(h)))
        self._h = h
        self._n_rows = 0

    # mapper::load, from memory
    def load_index(self, rows: np.ndarray, kmer_to_index: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        k2i = np.ascontiguousarray(kmer_to_index, dtype=np.int32)
        row_bytes = (self.params.num_buckets + 7) >> 3
        n_rows = rows.size // row_bytes if row_bytes else 0
        if rows.size != n_rows * row_bytes:
            raise ValueError("rows must hold whole rows of ceil(NB/8) bytes")
        _check(lib().bmf_load_index(self._h, _ptr(rows, _u8p), n_rows, _ptr(k2i, _i32p), k2i.size))
        self._n_rows = n_rows

    def load_index_ptr(self, rows_ptr, n_rows: int, k2i_ptr, n_kmers: int) -> None:
        """Same, from raw pointers (e.g. straight out of libbmhost, no numpy copy)."""
        _check(lib().bmf_load_index(self._h, C.cast(rows_ptr, _u8p), n_rows, C.cast(k2i_ptr, _i32p), n_kmers))
        self._n_rows = n_rows

    def build_index(self, genome_bytes, bucket_start, bucket_len, kmer_to_index) -> None:
        """GPU form of the host indexer: rows are built in HBM from the genome and the bucket views."""
        g = np.ascontiguousarray(genome_bytes, dtype=np.uint8)
        bs = np.ascontiguousarray(bucket_start, dtype=np.uint64)
        bl = np.ascontiguousarray(bucket_len, dtype=np.uint32)
        k2i = np.ascontiguousarray(kmer_to_index, dtype=np.int32)
        _check(lib().bmf_build_index(self._h, _ptr(g, _u8p), len(g), _ptr(bs, _u64p), _ptr(bl, _u32p), len(bs),
                                     _ptr(k2i, _i32p), len(k2i)))
        self._n_rows = int((k2i >= 0).sum())

    def index_download(self) -> np.ndarray:
        """The loaded index in the .qgram layout: n_rows x ceil(NB/8) bytes."""
        n = C.c_uint64()
        _check(lib().bmf_index_download(self._h, None, C.byref(n)))
        rows = np.zeros((n.value, (self.params.num_buckets + 7) >> 3), dtype=np.uint8)
        _check(lib().bmf_index_download(self._h, _ptr(rows, _u8p), C.byref(n)))
        return rows

    # mapper::load, from <dir>/<indicator>.{kmers_index,qgram}
    def load_index_files(self, index_dir: str, indicator: str) -> None:
        _check(lib().bmf_load_index_files(self._h, os.fsencode(index_dir), indicator.encode()))
        n = C.c_uint64()
        _check(lib().bmf_index_download(self._h, None, C.byref(n)))
        self._n_rows = int(n.value)

    # mapper::reset
    def reset(self) -> None:
        _check(lib().bmf_reset(self._h))
        self._n_rows = 0

    def zeros(self, n_rows: int | None = None) -> np.ndarray:
        n = self._n_rows if n_rows is None else n_rows
        out = np.zeros(n, dtype=np.uint32)
        _check(lib().bmf_index_zeros(self._h, _ptr(out, _u32p)))
        return out

    def map_windows(self, bases, quals, win_start, win_len, out=None):
        """query_sequence for every window; returns (counts[n,2], buckets[n,2,max_candidates]).  `out` = a pair of
        arrays of those shapes to write into (entries past a list's count are left as they are)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        win_start = np.ascontiguousarray(win_start, dtype=np.uint64)
        win_len = np.ascontiguousarray(win_len, dtype=np.uint32)
        n = len(win_start)
        assert len(win_len) == n and len(quals) == len(bases)
        mc = self.params.max_candidates
        if out is None:
            counts = np.zeros((n, 2), dtype=np.uint32)
            buckets = np.zeros((n, 2, mc), dtype=np.uint32)
        else:
            counts, buckets = out
            assert counts.shape == (n, 2) and buckets.shape == (n, 2, mc) and counts.dtype == buckets.dtype == np.uint32
            assert counts.flags.c_contiguous and buckets.flags.c_contiguous
        _check(lib().bmf_map_windows(self._h, _ptr(bases, _u8p), _ptr(quals, _u8p), len(bases),
                                     _ptr(win_start, _u64p), _ptr(win_len, _u32p), n,
                                     _ptr(counts, _u32p), _ptr(buckets, _u32p)))
        return counts, buckets

    def map_windows_compact(self, bases, quals, win_start, win_len, out=None):
        """Same, packed: (counts[n,2], ids) with the lists back to back in window order.  `out` = (counts, ids buffer)
        to reuse; the ids come back as a view of the buffer."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        win_start = np.ascontiguousarray(win_start, dtype=np.uint64)
        win_len = np.ascontiguousarray(win_len, dtype=np.uint32)
        n = len(win_start)
        if out is None:
            out = (np.zeros((n, 2), dtype=np.uint32), np.empty(2 * n * self.params.max_candidates, dtype=np.uint32))
        counts, ids = out
        used = C.c_uint64()
        _check(lib().bmf_map_windows_compact(self._h, _ptr(bases, _u8p), _ptr(quals, _u8p), len(bases), _ptr(win_start, _u64p),
                                             _ptr(win_len, _u32p), n, _ptr(counts, _u32p), _ptr(ids, _u32p), ids.size,
                                             C.byref(used)))
        return counts, ids[: used.value]

    def map_text_windows_compact(self, text, seq_start, qual_start, win_len, out=None):
        """bmf_map_text_windows_compact: windows whose bases and qualities lie apart in one buffer (a FASTQ text); the
        library gathers them.  Returns (counts[n,2], ids) as map_windows_compact."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        seq_start = np.ascontiguousarray(seq_start, dtype=np.uint64)
        qual_start = np.ascontiguousarray(qual_start, dtype=np.uint64)
        win_len = np.ascontiguousarray(win_len, dtype=np.uint32)
        n = len(seq_start)
        if out is None:
            out = (np.zeros((n, 2), dtype=np.uint32), np.empty(2 * n * self.params.max_candidates, dtype=np.uint32))
        counts, ids = out
        used = C.c_uint64()
        _check(lib().bmf_map_text_windows_compact(self._h, _ptr(text, _u8p), len(text), _ptr(seq_start, _u64p), _ptr(qual_start, _u64p),
                                                  _ptr(win_len, _u32p), n, _ptr(counts, _u32p), _ptr(ids, _u32p), ids.size,
                                                  C.byref(used)))
        return counts, ids[: used.value]

    def batch(self, bases, quals, win_start, win_len) -> Batch:
        return Batch(self, bases, quals, win_start, win_len)

    def sync(self) -> None:
        _check(lib().bmf_sync(self._h))

    def profile_begin(self, max_runs: int) -> None:
        _check(lib().bmf_profile_begin(self._h, max_runs))

    def profile_end(self, max_runs: int):
        n = C.c_uint32()
        a = np.zeros(max_runs, dtype=np.float32)
        b = np.zeros(max_runs, dtype=np.float32)
        _check(lib().bmf_profile_end(self._h, C.byref(n), _ptr(a, _f32p), _ptr(b, _f32p)))
        return a[: n.value].copy(), b[: n.value].copy()

    def info(self) -> dict:
        v = [C.c_uint32() for _ in range(4)]
        _check(lib().bmf_info(self._h, *[C.byref(x) for x in v]))
        r, fold, frows = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().bmf_pass1_rows(self._h, C.byref(r)))
        _check(lib().bmf_pass1_fold(self._h, C.byref(fold), C.byref(frows)))
        return {"row_pitch_bytes": v[0].value, "chunks_per_lane": v[1].value, "planes": v[2].value,
                "rows_in_flight": v[3].value, "pass1_rows": r.value, "pass1_fold": fold.value, "pass1_fold_rows": frows.value}

    def close(self) -> None:
        if self._h:
            lib().bmf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
