"""Second, independent formulation of the candidate-bucket vote (numpy).  TEST INFRASTRUCTURE ONLY.

The C oracle (bm_oracle.c) applies the reference's F unary bit-plane update literally
(bucket_map/mapper/q_gram_mapper.h:75-102).  This module states the same function the other way:
count, per bucket, how many samples missed, and return the argmin set when the minimum is below F
(SURVEY.md 3.3).  tests/test_oracle.py checks that the two agree on random inputs -- the substitute
for the golden vectors the reference does not have ("parity unpinned", see bm_oracle.h).

Pure numpy / Python loops: only for small cases.
"""
from __future__ import annotations

import math

import numpy as np

_RANK = np.zeros(256, dtype=np.uint8)
for _chars, _r in (("AaRrWwMmDdHhVv", 0), ("CcYySsBb", 1), ("GgKk", 2), ("TtUu", 3)):
    for _c in _chars:
        _RANK[ord(_c)] = _r


def dna4_ranks(bases) -> np.ndarray:
    """SeqAn3 dna4 assign_char folding."""
    b = np.frombuffer(bases, np.uint8) if isinstance(bases, (bytes, bytearray)) else np.asarray(bases, np.uint8)
    return _RANK[b]


def kmer_hashes(bases, k: int) -> np.ndarray:
    """views::kmer_hash(ungapped{k}): big-endian base 4, A,C,G,T = 0..3 (App. C.1)."""
    r = dna4_ranks(bases).astype(np.uint64)
    n = len(r) - k + 1
    if k <= 0 or n <= 0:
        return np.zeros(0, np.uint32)
    h = np.zeros(n, np.uint64)
    for t in range(k):
        h = h * 4 + r[t:t + n]
    return h.astype(np.uint32)


def kmer_qualities(quals, k: int) -> np.ndarray:
    """quality_filter.h:611-631: sliding sum of phred ranks over k bases."""
    q = (np.frombuffer(quals, np.uint8) if isinstance(quals, (bytes, bytearray)) else np.asarray(quals, np.uint8))
    q = q.astype(np.int64) - 33
    n = len(q) - k + 1
    if k <= 0 or n <= 0:
        return np.zeros(0, np.uint32)
    c = np.concatenate(([0], np.cumsum(q)))
    return (c[k:] - c[:-k]).astype(np.uint32)


def sample_positions(n: int, upper_bound: int) -> list[int]:
    """utils.h:160-178 (stateless; upper_bound == 0 -> zeros, see bm_oracle.h)."""
    if n == 0:
        return []
    delta = 0.0 if n == 1 else float(upper_bound + 1) / (n - 1)
    return [int(math.floor(i * delta)) for i in range(n - 1)] + [upper_bound]


def hash_reverse_complement(h: int, k: int) -> int:
    """utils.h:291-302, stated as: complement every base, reverse the base order."""
    bases = [(h >> (2 * (k - 1 - i))) & 3 for i in range(k)]          # first base first
    rc = [3 - b for b in reversed(bases)]
    out = 0
    for b in rc:
        out = (out << 2) | b
    return out


def unpack_rows(rows: np.ndarray, num_buckets: int) -> np.ndarray:
    """.qgram rows (n_rows x ceil(NB/8) bytes, LSB-first) -> bool matrix n_rows x NB."""
    rows = np.asarray(rows, np.uint8)
    row_bytes = (num_buckets + 7) >> 3
    rows = rows.reshape(-1, row_bytes)
    return np.unpackbits(rows, axis=1, bitorder="little")[:, :num_buckets].astype(bool)


def query_miss_counts(bits: np.ndarray, kmer_to_index: np.ndarray, hashes, *, k: int, q: int, num_fault: int):
    """Integer formulation of q_gram_mapper::query: returns (ascending argmin buckets, miss counts)."""
    nb = bits.shape[1]
    misses = np.zeros(nb, np.int64)
    if bits.shape[0] == 0:
        return np.zeros(0, np.uint32), misses
    qmask = (1 << (2 * q)) - 1
    for h in hashes:
        present = np.ones(nb, bool)
        for i in range(k - q + 1):
            g = (int(h) >> (2 * i)) & qmask
            idx = int(kmer_to_index[g]) if g < len(kmer_to_index) else -1
            if idx >= 0:
                present &= bits[idx]
        misses += ~present
    m = int(misses.min())
    if m > num_fault - 1:
        return np.zeros(0, np.uint32), misses
    return np.nonzero(misses == m)[0].astype(np.uint32), misses


def query_sequence(bits, kmer_to_index, zeros, bases, quals, *, k, q, num_samples, num_fault, threshold,
                   min_base_quality, max_candidates=30):
    """q_gram_mapper::query_sequence (q_gram_mapper.h:414-480) in the integer formulation."""
    hs = kmer_hashes(bases, k)
    qs = kmer_qualities(quals, k)
    qmask = (1 << (2 * q)) - 1
    good = []
    for h, s in zip(hs, qs):
        dist = False
        for i in range(k - q + 1):
            g = (int(h) >> (2 * i)) & qmask
            idx = int(kmer_to_index[g]) if g < len(kmer_to_index) else -1
            if idx >= 0 and zeros[idx] >= threshold:
                dist = True
                break
        if dist and s >= min_base_quality:
            good.append(int(h))
    empty = np.zeros(0, np.uint32)
    if len(good) < 0.2 * num_samples:
        return empty, empty
    pos = sample_positions(num_samples, len(good) - 1)
    smp = [good[p] for p in pos]
    fwd, _ = query_miss_counts(bits, kmer_to_index, smp, k=k, q=q, num_fault=num_fault)
    rc, _ = query_miss_counts(bits, kmer_to_index, [hash_reverse_complement(h, k) for h in smp], k=k, q=q,
                              num_fault=num_fault)
    if len(fwd) > max_candidates:
        fwd = empty
    if len(rc) > max_candidates:
        rc = empty
    return fwd, rc
