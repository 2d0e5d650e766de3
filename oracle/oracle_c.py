"""ctypes wrapper of the C oracle (oracle/libbm_oracle.so).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/bm_oracle.h): the reference cannot be built here and ships no golden
vectors, so this oracle is pinned by hand-derived known-answer vectors and a dual-formulation
property test only.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product (bucket-map_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbm_oracle.so")


class Params(C.Structure):
    _fields_ = [
        ("num_buckets", C.c_uint32), ("q", C.c_uint32), ("k", C.c_uint32), ("num_samples", C.c_uint32),
        ("num_fault", C.c_uint32), ("threshold", C.c_uint32), ("min_base_quality", C.c_uint32),
        ("max_candidates", C.c_uint32), ("read_len", C.c_uint32), ("num_segment_samples", C.c_uint32),
    ]


class LocParams(C.Structure):
    """bmlo_params (oracle/bm_locator_oracle.h)."""
    _fields_ = [("k", C.c_uint32), ("num_samples", C.c_uint32), ("allowed_mismatch", C.c_int32),
                ("allowed_indel", C.c_int32)]


_u8p, _u32p, _u64p, _i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32))
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make oracle/libbm_oracle.so`")
        L = C.CDLL(LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        sig = {
            "bmo_fault_from_rate": (u32, [u32, C.c_float]),
            "bmo_threshold": (u32, [C.c_float, u32]),
            "bmo_ceil_mul_f32": (u32, [C.c_float, u32]),
            "bmo_sample_positions": (None, [u32, u32, _u32p]),
            "bmo_hash_reverse_complement": (u32, [u32, u32]),
            "bmo_dna4_rank": (C.c_uint8, [C.c_uint8]),
            "bmo_kmer_hashes": (u32, [_u8p, u32, u32, _u32p]),
            "bmo_kmer_qualities": (u32, [_u8p, u32, u32, _u32p]),
            "bmo_window_starts": (u32, [u32, u32, u32, _u32p]),
            "bmo_index_create": (vp, [C.POINTER(Params), _u8p, C.c_uint64, _i32p, C.c_uint64]),
            "bmo_index_load": (vp, [C.POINTER(Params), C.c_char_p, C.c_char_p]),
            "bmo_index_destroy": (None, [vp]),
            "bmo_index_rows": (C.c_uint64, [vp]),
            "bmo_index_zeros": (vp, [vp]),
            "bmo_is_highly_distinguishable": (C.c_int, [vp, u32]),
            "bmo_query": (u32, [vp, _u32p, u32, _u32p]),
            "bmo_query_sequence": (None, [vp, _u8p, _u8p, u32, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p]),
            "bmo_map_windows": (C.c_uint64, [vp, _u8p, _u8p, _u64p, _u32p, u32, _u32p, _u32p]),
            "bmlo_locate": (C.c_int, [C.POINTER(LocParams), _u8p, _u64p, _u32p, u32, _u32p, C.POINTER(C.c_uint16), _u32p,
                                      _u32p, _u32p, _u8p, u32, C.POINTER(C.c_int32), _u32p]),
            "bmlo_sample_windows": (None, [u32, u32, u32, _u8p, _u8p, _u64p, _u32p, u32, _u32p, C.POINTER(C.c_uint16), _u8p]),
            "bmao_align": (C.c_int, [_u8p, u32, C.c_int, _u8p, u32, _i32p, _u32p, _u32p, u32]),
            "bmao_align_batch": (C.c_int, [_u8p, _u8p, _u64p, _u32p, _u8p, _u64p, _u32p, u32, _i32p, _u32p, _u64p, _u32p,
                                           C.c_uint64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ty)


def make_params(num_buckets, q=9, k=12, num_samples=15, num_fault=6, threshold=0, min_base_quality=300,
                max_candidates=30, read_len=300, num_segment_samples=5) -> Params:
    return Params(num_buckets, q, k, num_samples, num_fault, threshold, min_base_quality, max_candidates, read_len,
                  num_segment_samples)


def params_from_cli(num_buckets, *, index_seed=9, query_seed=12, read_len=300, mapper_samples=15,
                    max_error_rate=0.4, distinguishability=0.5, average_base_quality=25) -> Params:
    L = lib()
    return make_params(num_buckets, index_seed, query_seed, mapper_samples,
                       L.bmo_fault_from_rate(mapper_samples, max_error_rate),
                       L.bmo_threshold(distinguishability, num_buckets), average_base_quality * query_seed, 30,
                       read_len, 5)


def sample_positions(n: int, upper_bound: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint32)
    lib().bmo_sample_positions(n, upper_bound, _p(out, _u32p))
    return out


def kmer_hashes(bases: bytes | np.ndarray, k: int) -> np.ndarray:
    b = np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, np.uint8)
    out = np.zeros(max(len(b), 1), dtype=np.uint32)
    n = lib().bmo_kmer_hashes(_p(b, _u8p), len(b), k, _p(out, _u32p))
    return out[:n].copy()


def kmer_qualities(quals: bytes | np.ndarray, k: int) -> np.ndarray:
    b = np.frombuffer(quals, dtype=np.uint8) if isinstance(quals, (bytes, bytearray)) else np.ascontiguousarray(quals, np.uint8)
    out = np.zeros(max(len(b), 1), dtype=np.uint32)
    n = lib().bmo_kmer_qualities(_p(b, _u8p), len(b), k, _p(out, _u32p))
    return out[:n].copy()


def window_starts(record_len: int, read_len: int, n_seg: int = 5) -> np.ndarray:
    out = np.zeros(max(n_seg, 1), dtype=np.uint32)
    n = lib().bmo_window_starts(record_len, read_len, n_seg, _p(out, _u32p))
    return out[:n].copy()


def locate(k, num_samples, allowed_mismatch, allowed_indel, genome, bucket_start, bucket_len, sample_hash, sample_pos,
           seg_len, pair_bucket, pair_window, pair_rc):
    """bmlo_locate: _create_kmer_index + _find_offset for every candidate (bucket_locator.h:162-177,209-290)."""
    prm = LocParams(k, num_samples, allowed_mismatch, allowed_indel)
    g = np.ascontiguousarray(genome, np.uint8)
    bs, bl = np.ascontiguousarray(bucket_start, np.uint64), np.ascontiguousarray(bucket_len, np.uint32)
    sh, sp = np.ascontiguousarray(sample_hash, np.uint32), np.ascontiguousarray(sample_pos, np.uint16)
    sl = np.ascontiguousarray(seg_len, np.uint32)
    pb, pw = np.ascontiguousarray(pair_bucket, np.uint32), np.ascontiguousarray(pair_window, np.uint32)
    pr = np.ascontiguousarray(pair_rc, np.uint8)
    off = np.full(len(pb), -1, np.int32)
    votes = np.zeros(len(pb), np.uint32)
    rc = lib().bmlo_locate(C.byref(prm), _p(g, _u8p), _p(bs, _u64p), _p(bl, _u32p), len(bs), _p(sh, _u32p),
                           _p(sp, C.POINTER(C.c_uint16)), _p(sl, _u32p), _p(pb, _u32p), _p(pw, _u32p), _p(pr, _u8p),
                           len(pb), _p(off, C.POINTER(C.c_int32)), _p(votes, _u32p))
    if rc:
        raise RuntimeError("oracle locator: bad bucket id")
    return off, votes


def sample_windows(k, p, min_base_quality, bases, quals, win_start, win_len):
    """bmlo_sample_windows: _prepare_read_query's sampling (bucket_locator.h:292-347)."""
    b, q = np.ascontiguousarray(bases, np.uint8), np.ascontiguousarray(quals, np.uint8)
    ws, wl = np.ascontiguousarray(win_start, np.uint64), np.ascontiguousarray(win_len, np.uint32)
    n = len(ws)
    h = np.zeros((n, p), np.uint32)
    pos = np.zeros((n, p), np.uint16)
    has = np.zeros(n, np.uint8)
    lib().bmlo_sample_windows(k, p, min_base_quality, _p(b, _u8p), _p(q, _u8p), _p(ws, _u64p), _p(wl, _u32p), n,
                              _p(h, _u32p), _p(pos, C.POINTER(C.c_uint16)), _p(has, _u8p))
    return h, pos, has


def align(text: bytes, query: bytes, text_rc: bool = False):
    """bmao_align: (score, begin, CIGAR string) of the whole query against the best substring of the text
    (bucket_locator.h:520-528,569-576; tie rules in oracle/bm_align_oracle.h)."""
    t = np.frombuffer(bytes(text), np.uint8) if len(text) else np.zeros(1, np.uint8)
    q = np.frombuffer(bytes(query), np.uint8) if len(query) else np.zeros(1, np.uint8)
    cap = len(text) + len(query) + 1
    cg = np.zeros(cap, np.uint32)
    score, begin = C.c_int32(), C.c_uint32()
    n = lib().bmao_align(_p(t, _u8p), len(text), int(text_rc), _p(q, _u8p), len(query), C.byref(score), C.byref(begin),
                         _p(cg, _u32p), cap)
    if n < 0:
        raise MemoryError("bmao_align")
    return score.value, begin.value, cigar_string(cg[:n])


def cigar_string(packed) -> str:
    return "".join(f"{int(e) >> 4}{'MID'[int(e) & 15]}" for e in packed)


def align_batch(genome, reads, text_start, text_len, text_rc, query_start, query_len):
    """bmao_align_batch, buffer layout of bmv_align (include/bmv.h)."""
    g = np.ascontiguousarray(genome, np.uint8)
    r = np.ascontiguousarray(reads, np.uint8)
    ts, tl = np.ascontiguousarray(text_start, np.uint64), np.ascontiguousarray(text_len, np.uint32)
    trc = np.ascontiguousarray(text_rc, np.uint8)
    qs, ql = np.ascontiguousarray(query_start, np.uint64), np.ascontiguousarray(query_len, np.uint32)
    n = len(ts)
    score, begin = np.zeros(n, np.int32), np.zeros(n, np.uint32)
    off = np.zeros(n + 1, np.uint64)
    cap = int(tl.astype(np.uint64).sum() + ql.astype(np.uint64).sum()) + n + 1
    cg = np.zeros(cap, np.uint32)
    rc = lib().bmao_align_batch(_p(g, _u8p), _p(r, _u8p), _p(ts, _u64p), _p(tl, _u32p), _p(trc, _u8p), _p(qs, _u64p),
                                _p(ql, _u32p), n, _p(score, _i32p), _p(begin, _u32p), _p(off, _u64p), _p(cg, _u32p), cap)
    if rc:
        raise RuntimeError("oracle verifier failed")
    return score, begin, off, cg[: int(off[n])].copy()


def check_alignments(genome, reads, text_start, text_len, text_rc, query_start, query_len, score, begin, cigar_offset, cigar,
                     threads=None):
    """bmao_check_batch over host threads: per alignment 0 = the score is the two-row DP's optimum AND the CIGAR is a valid
    path of exactly that cost; else a bit set (1 score, 2 CIGAR shape, 4 CIGAR cost, 8 memory)."""
    from concurrent.futures import ThreadPoolExecutor
    g, r = np.ascontiguousarray(genome, np.uint8), np.ascontiguousarray(reads, np.uint8)
    ts, tl = np.ascontiguousarray(text_start, np.uint64), np.ascontiguousarray(text_len, np.uint32)
    trc = np.ascontiguousarray(text_rc, np.uint8)
    qs, ql = np.ascontiguousarray(query_start, np.uint64), np.ascontiguousarray(query_len, np.uint32)
    sc, bg = np.ascontiguousarray(score, np.int32), np.ascontiguousarray(begin, np.uint32)
    co, cg = np.ascontiguousarray(cigar_offset, np.uint64), np.ascontiguousarray(cigar if len(cigar) else np.zeros(1), np.uint32)
    n = len(ts)
    bad = np.full(n, 255, np.uint8)
    threads = threads or max(1, min(len(os.sched_getaffinity(0)), 32))
    fn = lib().bmao_check_batch
    fn.restype = None
    fn.argtypes = [_u8p, _u8p, _u64p, _u32p, _u8p, _u64p, _u32p, C.c_uint32, C.c_uint32, _i32p, _u32p, _u64p, _u32p, _u8p]
    # cut by cells, not by count: the alignments of a batch differ a hundredfold in size
    cells = np.cumsum(tl.astype(np.float64) * ql.astype(np.float64))
    cuts = [0] + [int(np.searchsorted(cells, cells[-1] * (t + 1) / (4 * threads))) for t in range(4 * threads - 1)] + [n] if n else [0, 0]
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda ab: fn(_p(g, _u8p), _p(r, _u8p), _p(ts, _u64p), _p(tl, _u32p), _p(trc, _u8p), _p(qs, _u64p),
                                    _p(ql, _u32p), ab[0], ab[1], _p(sc, _i32p), _p(bg, _u32p), _p(co, _u64p), _p(cg, _u32p),
                                    _p(bad, _u8p)) if ab[1] > ab[0] else None, zip(cuts[:-1], cuts[1:])))
    return bad


class Index:
    def __init__(self, params: Params, rows: np.ndarray | None = None, kmer_to_index: np.ndarray | None = None, *,
                 rows_ptr=None, n_rows=None, k2i_ptr=None, n_kmers=None, files=None):
        self.params = params
        if files is not None:
            self._h = lib().bmo_index_load(C.byref(params), os.fsencode(files[0]), files[1].encode())
        elif rows_ptr is not None:
            self._h = lib().bmo_index_create(C.byref(params), C.cast(rows_ptr, _u8p), n_rows, C.cast(k2i_ptr, _i32p), n_kmers)
        else:
            rows = np.ascontiguousarray(rows, dtype=np.uint8)
            k2i = np.ascontiguousarray(kmer_to_index, dtype=np.int32)
            row_bytes = (params.num_buckets + 7) >> 3
            self._h = lib().bmo_index_create(C.byref(params), _p(rows, _u8p), rows.size // row_bytes, _p(k2i, _i32p), k2i.size)
        if not self._h:
            raise RuntimeError("oracle: cannot create index")

    @property
    def n_rows(self) -> int:
        return lib().bmo_index_rows(self._h)

    def zeros(self) -> np.ndarray:
        n = self.n_rows
        ptr = lib().bmo_index_zeros(self._h)
        return np.frombuffer((C.c_uint32 * n).from_address(ptr), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32)

    def is_highly_distinguishable(self, h: int) -> bool:
        return bool(lib().bmo_is_highly_distinguishable(self._h, h))

    def query(self, hashes) -> np.ndarray:
        hs = np.ascontiguousarray(hashes, dtype=np.uint32)
        out = np.zeros(self.params.num_buckets, dtype=np.uint32)
        n = lib().bmo_query(self._h, _p(hs, _u32p), len(hs), _p(out, _u32p))
        return out[:n].copy()

    def query_sequence(self, bases, quals):
        b = np.ascontiguousarray(np.frombuffer(bases, np.uint8) if isinstance(bases, (bytes, bytearray)) else bases, np.uint8)
        q = np.ascontiguousarray(np.frombuffer(quals, np.uint8) if isinstance(quals, (bytes, bytearray)) else quals, np.uint8)
        mc, S = self.params.max_candidates, self.params.num_samples
        of, orc = np.zeros(mc, np.uint32), np.zeros(mc, np.uint32)
        nf, nr, ng = C.c_uint32(), C.c_uint32(), C.c_uint32()
        smp = np.zeros(S, np.uint32)
        lib().bmo_query_sequence(self._h, _p(b, _u8p), _p(q, _u8p), len(b), _p(of, _u32p), C.byref(nf), _p(orc, _u32p),
                                 C.byref(nr), _p(smp, _u32p), C.byref(ng))
        return of[: nf.value].copy(), orc[: nr.value].copy(), smp, ng.value

    def map_windows(self, bases, quals, win_start, win_len):
        b = np.ascontiguousarray(bases, np.uint8)
        q = np.ascontiguousarray(quals, np.uint8)
        ws = np.ascontiguousarray(win_start, np.uint64)
        wl = np.ascontiguousarray(win_len, np.uint32)
        n = len(ws)
        assert len(wl) == n
        mc = self.params.max_candidates
        counts = np.zeros((n, 2), np.uint32)
        buckets = np.zeros((n, 2, mc), np.uint32)
        rows = lib().bmo_map_windows(self._h, _p(b, _u8p), _p(q, _u8p), _p(ws, _u64p), _p(wl, _u32p), n,
                                     _p(counts, _u32p), _p(buckets, _u32p))
        return counts, buckets, int(rows)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bmo_index_destroy(self._h)
            self._h = None
