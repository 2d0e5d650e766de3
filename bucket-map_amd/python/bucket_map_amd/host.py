"""ctypes binding of the host plumbing (libbmhost.so): synthetic genomes, FASTA I/O, bucket cutting,
the host indexer and the read simulator.  Inputs for tests and bench.py; no GPU code and no part of
the filter's arithmetic lives here."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _PKG_ROOT

LIBBMHOST_PATH = os.path.join(_PKG_ROOT, "libbmhost.so")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIBBMHOST_PATH):
            raise ImportError(f"{LIBBMHOST_PATH} is missing: run `make`")
        L = C.CDLL(LIBBMHOST_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        sig = {
            "bmh_last_error": (C.c_char_p, []),
            "bmh_genome_synth": (vp, [u64, C.POINTER(u64), u32, u32]),
            "bmh_genome_synth_skewed": (vp, [u64, C.POINTER(u64), u32, u32, C.c_double, C.c_double]),
            "bmh_genome_gap_bases": (u64, [vp]),
            "bmh_genome_read_fasta": (vp, [C.c_char_p]),
            "bmh_genome_write_fasta": (C.c_int, [vp, C.c_char_p]),
            "bmh_genome_free": (None, [vp]),
            "bmh_genome_records": (u32, [vp]),
            "bmh_genome_record_len": (u64, [vp, u32]),
            "bmh_genome_record_id": (C.c_char_p, [vp, u32]),
            "bmh_genome_record_seq": (vp, [vp, u32]),
            "bmh_genome_total": (u64, [vp]),
            "bmh_genome_flatten": (None, [vp, vp, vp]),
            "bmh_select_qgrams": (u64, [u32, C.c_float, u64, vp]),
            "bmh_select_qgrams_xy": (u64, [u32, C.c_float, u64, u64, u64, u64, vp]),
            "bmh_fastq_stats": (C.c_int, [C.c_char_p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
            "bmh_awk_bucket_num": (u32, [vp, u32]),
            "bmh_cut_buckets": (u32, [vp, u32, u32, C.POINTER(u32)]),
            "bmh_index_build": (vp, [vp, u32, u32, u32, u32, C.c_float, u64, u32]),
            "bmh_index_free": (None, [vp]),
            "bmh_index_num_rows": (u64, [vp]),
            "bmh_index_row_bytes": (u32, [vp]),
            "bmh_index_rows": (vp, [vp]),
            "bmh_index_kmer_to_index": (vp, [vp]),
            "bmh_index_num_kmers": (u64, [vp]),
            "bmh_index_write": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
            "bmh_reads_simulate": (vp, [vp, u32, u32, u32, u64, C.c_double, C.c_double, C.c_double, u64, u32, u32]),
            "bmh_reads_free": (None, [vp]),
            "bmh_reads_count": (u64, [vp]),
            "bmh_reads_bases": (vp, [vp]),
            "bmh_reads_quals": (vp, [vp]),
            "bmh_reads_offsets": (vp, [vp]),
            "bmh_reads_truth_bucket": (vp, [vp]),
            "bmh_reads_truth_offset": (vp, [vp]),
            "bmh_reads_truth_rc": (vp, [vp]),
            "bmh_reads_write_fastq": (C.c_int, [vp, vp, u32, u32, C.c_char_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _err() -> str:
    return lib().bmh_last_error().decode(errors="replace")


def _view(ptr, n, dtype) -> np.ndarray:
    """numpy view (no copy) of n items at ptr; the owner object must stay alive."""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


def select_qgrams(q: int, kmer_frac: float = 1.0, hash_seed: int = 20240004) -> np.ndarray:
    """FracMinHash row selection (bucket_indexer.h:147-157): kmer_to_index with 4^q entries."""
    out = np.zeros(4 ** q, dtype=np.int32)
    lib().bmh_select_qgrams(q, kmer_frac, hash_seed, out.ctypes.data)
    return out


def select_qgrams_xy(q: int, kmer_frac: float, x: int, y: int, p: int = 116731, table: int = 10000) -> np.ndarray:
    """The same selection with the hash (x*i + y) % p % table given outright (hash_function_generator.h:105-116)."""
    out = np.zeros(4 ** q, dtype=np.int32)
    lib().bmh_select_qgrams_xy(q, kmer_frac, x, y, p, table, out.ctypes.data)
    return out


def fastq_stats(path: str):
    """(records, bases, checksum over ids/sequences/qualities) as the tool's FASTQ reader sees the file."""
    n, b, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
    if lib().bmh_fastq_stats(os.fsencode(path), C.byref(n), C.byref(b), C.byref(h)):
        raise RuntimeError(_err())
    return n.value, b.value, h.value


class Genome:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError(_err())
        self._h = handle

    @classmethod
    def synth(cls, seed: int, record_lengths, threads: int = 0, profile: str = "uniform", sigma: float = 0.0,
              p_repeat: float = -1.0) -> "Genome":
        """profile "uniform": i.i.d. bases; "genome": the skewed, repetitive generator of bm_synth.h."""
        lens = (C.c_uint64 * len(record_lengths))(*[int(x) for x in record_lengths])
        if profile == "genome":
            return cls(lib().bmh_genome_synth_skewed(seed, lens, len(record_lengths), threads, sigma, p_repeat))
        if profile != "uniform":
            raise ValueError(f"unknown genome profile {profile!r}")
        return cls(lib().bmh_genome_synth(seed, lens, len(record_lengths), threads))

    def gap_bases(self) -> int:
        return lib().bmh_genome_gap_bases(self._h)

    @classmethod
    def read_fasta(cls, path: str) -> "Genome":
        return cls(lib().bmh_genome_read_fasta(os.fsencode(path)))

    def write_fasta(self, path: str) -> None:
        if lib().bmh_genome_write_fasta(self._h, os.fsencode(path)):
            raise RuntimeError(_err())

    @property
    def n_records(self) -> int:
        return lib().bmh_genome_records(self._h)

    def record_len(self, i: int) -> int:
        return lib().bmh_genome_record_len(self._h, i)

    def record_id(self, i: int) -> str:
        return lib().bmh_genome_record_id(self._h, i).decode()

    def record_seq(self, i: int) -> np.ndarray:
        return _view(lib().bmh_genome_record_seq(self._h, i), self.record_len(i), np.uint8)

    def total_length(self) -> int:
        return sum(self.record_len(i) for i in range(self.n_records))

    def flat(self):
        """(all records back to back as one uint8 array, record offsets[n_records+1])."""
        out = np.empty(lib().bmh_genome_total(self._h), dtype=np.uint8)
        off = np.zeros(self.n_records + 1, dtype=np.uint64)
        lib().bmh_genome_flatten(self._h, out.ctypes.data, off.ctypes.data)
        return out, off

    def bucket_views(self, bucket_len: int, read_len: int):
        """Kept buckets as (start, length) views into flat(): what bmf_build_index / bml_load_genome take."""
        b = self.cut_buckets(bucket_len, read_len)
        off = np.concatenate(([0], np.cumsum([self.record_len(i) for i in range(self.n_records)]))).astype(np.uint64)
        start = off[b[:, 0]] + b[:, 2].astype(np.uint64)
        return start.astype(np.uint64), (b[:, 3] - b[:, 2]).astype(np.uint32)

    def awk_bucket_num(self, bucket_len: int) -> int:
        return lib().bmh_awk_bucket_num(self._h, bucket_len)

    def cut_buckets(self, bucket_len: int, read_len: int) -> np.ndarray:
        n = lib().bmh_cut_buckets(self._h, bucket_len, read_len, None)
        out = np.zeros((n, 4), dtype=np.uint32)
        lib().bmh_cut_buckets(self._h, bucket_len, read_len, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bmh_genome_free(self._h)
            self._h = None


class Index:
    def __init__(self, genome: Genome, num_buckets: int, bucket_len: int, read_len: int, q: int = 9,
                 kmer_frac: float = 1.0, hash_seed: int = 20240004, threads: int = 0):
        self._h = lib().bmh_index_build(genome._h, num_buckets, bucket_len, read_len, q, kmer_frac, hash_seed, threads)
        if not self._h:
            raise RuntimeError(_err())
        self.num_buckets = num_buckets

    @property
    def num_rows(self) -> int:
        return lib().bmh_index_num_rows(self._h)

    @property
    def row_bytes(self) -> int:
        return lib().bmh_index_row_bytes(self._h)

    @property
    def rows_ptr(self):
        return lib().bmh_index_rows(self._h)

    @property
    def k2i_ptr(self):
        return lib().bmh_index_kmer_to_index(self._h)

    @property
    def num_kmers(self) -> int:
        return lib().bmh_index_num_kmers(self._h)

    def rows(self) -> np.ndarray:
        return _view(self.rows_ptr, self.num_rows * self.row_bytes, np.uint8).reshape(self.num_rows, self.row_bytes)

    def kmer_to_index(self) -> np.ndarray:
        return _view(self.k2i_ptr, self.num_kmers, np.int32)

    def write(self, directory: str, indicator: str) -> None:
        if lib().bmh_index_write(self._h, os.fsencode(directory), indicator.encode()):
            raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bmh_index_free(self._h)
            self._h = None


class Reads:
    def __init__(self, genome: Genome, bucket_len: int, index_read_len: int, read_len: int, n_reads: int,
                 sub: float = 0.002, ins: float = 0.00025, dele: float = 0.00025, seed: int = 20240003,
                 noisy_quals: bool = False, threads: int = 0):
        self._genome = genome
        self._geom = (bucket_len, index_read_len)
        self._h = lib().bmh_reads_simulate(genome._h, bucket_len, index_read_len, read_len, n_reads, sub, ins, dele,
                                           seed, 1 if noisy_quals else 0, threads)
        if not self._h:
            raise RuntimeError(_err())
        n = lib().bmh_reads_count(self._h)
        self.n = n
        self.offsets = _view(lib().bmh_reads_offsets(self._h), n + 1, np.uint64)
        total = int(self.offsets[-1]) if n else 0
        self.bases = _view(lib().bmh_reads_bases(self._h), total, np.uint8)
        self.quals = _view(lib().bmh_reads_quals(self._h), total, np.uint8)
        self.truth_bucket = _view(lib().bmh_reads_truth_bucket(self._h), n, np.uint32)
        self.truth_offset = _view(lib().bmh_reads_truth_offset(self._h), n, np.uint32)
        self.truth_rc = _view(lib().bmh_reads_truth_rc(self._h), n, np.uint8)

    def write_fastq(self, prefix: str) -> None:
        if lib().bmh_reads_write_fastq(self._h, self._genome._h, self._geom[0], self._geom[1], os.fsencode(prefix)):
            raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bmh_reads_free(self._h)
            self._h = None
