"""Generates the committed fixtures tests/golden/*.json.

The reference ships no golden vectors for this path and cannot be run here, so these vectors are
produced by the NUMPY formulation of the algorithm (oracle/bm_oracle_np.py: integer miss counts +
argmin), with the index built by a brute-force numpy indexer written here.  They pin the C oracle
(which applies the reference's bit-plane update literally) and, through it, the HIP path.

    python tests/golden/make_golden.py        # rewrites tiny_index.json and reads_small.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import bm_oracle_np as onp  # noqa: E402


def tiny_index():
    rng = np.random.default_rng(20240010)
    nb, q, k = 70, 2, 3
    k2i = np.full(16, -1, np.int32)
    kept = [0, 1, 2, 4, 5, 7, 8, 9, 11, 12, 14, 15]          # 4 q-grams are not indexed
    k2i[kept] = np.arange(len(kept))
    bits = rng.random((len(kept), nb)) < 0.35
    bits[:, 66:] = False                                        # padding buckets: never set in any row
    rows = np.packbits(bits, axis=1, bitorder="little")
    cases = []
    for S, F, hashes in (
        (1, 1, [[h] for h in range(64)]),                       # every k-mer, exhaustively
        (2, 1, [[a, b] for a in range(64) for b in range(0, 64, 3)]),
        (2, 2, [[a, b] for a in range(0, 64, 2) for b in range(64)]),
        (3, 2, [list(map(int, rng.integers(0, 64, 3))) for _ in range(300)]),
        (5, 3, [list(map(int, rng.integers(0, 64, 5))) for _ in range(300)]),
    ):
        exp = [list(map(int, onp.query_miss_counts(bits, k2i, hs, k=k, q=q, num_fault=F)[0])) for hs in hashes]
        cases.append({"S": S, "F": F, "hashes": hashes, "expected": exp})
    return {"num_buckets": nb, "q": q, "k": k, "rows": rows.tolist(), "kmer_to_index": k2i.tolist(), "cases": cases}


def reads_small():
    rng = np.random.default_rng(20240011)
    q, k, bucket_len, read_len = 5, 7, 256, 60
    genome = rng.integers(0, 4, 9100).astype(np.uint8)
    letters = np.frombuffer(b"ACGT", np.uint8)
    # bucket cutting as utils.h:72-97 (single record)
    n = int(np.ceil(np.float32(len(genome)) / np.float32(bucket_len)))
    buckets = []
    for i in range(n):
        s, e = i * bucket_len, min(i * bucket_len + bucket_len + read_len, len(genome))
        if e - s > read_len:
            buckets.append((s, e))
    nb = n + 1                                                   # one padding bucket, like NB > #buckets
    k2i = np.arange(4 ** q, dtype=np.int32)
    k2i[rng.random(4 ** q) < 0.2] = -1                           # FracMinHash-like holes
    k2i[k2i >= 0] = np.arange((k2i >= 0).sum())
    bits = np.zeros((int((k2i >= 0).sum()), nb), bool)
    for b, (s, e) in enumerate(buckets):
        h = onp.kmer_hashes(letters[genome[s:e]], q)
        idx = k2i[h]
        bits[idx[idx >= 0], b] = True
    rows = np.packbits(bits, axis=1, bitorder="little")
    zeros = nb - bits.sum(axis=1)
    S, F, threshold, minq = 8, 4, int(np.float32(0.5) * np.float32(nb)), 20 * k
    reads = []
    for r in range(40):
        b = int(rng.integers(0, len(buckets)))
        s, e = buckets[b]
        start = int(rng.integers(0, e - s - read_len - 1))
        seq = genome[s + start: s + start + read_len].copy()
        for _ in range(int(rng.poisson(0.8))):
            seq[int(rng.integers(0, len(seq)))] = rng.integers(0, 4)
        text = letters[seq]
        rc = bool(rng.integers(0, 2))
        if rc:
            text = letters[3 - seq[::-1]]
        quals = np.full(len(text), ord("E"), np.uint8)
        if r % 5 == 1:
            quals = rng.integers(33, 33 + 42, len(text)).astype(np.uint8)   # noisy: quality filter bites
        if r % 7 == 2:
            text = text.copy(); text[10:14] = ord("N")                      # ambiguity codes fold to A
        if r == 3:
            text, quals = text[:5], quals[:5]                                # shorter than k
        if r == 4:
            quals[:] = ord("#")                                              # all low quality -> rejected
        if r == 6:
            text, quals = text[:0], quals[:0]                                # empty
        fwd, rcs = onp.query_sequence(bits, k2i, zeros, text, quals, k=k, q=q, num_samples=S, num_fault=F,
                                      threshold=threshold, min_base_quality=minq)
        reads.append({"bases": bytes(text).decode(), "quals": bytes(quals).decode(), "fwd": list(map(int, fwd)),
                      "rc": list(map(int, rcs)), "truth_bucket": b, "truth_rc": rc})
    return {"num_buckets": nb, "q": q, "k": k, "S": S, "F": F, "threshold": threshold, "min_base_quality": minq,
            "read_len": read_len, "rows": rows.tolist(), "kmer_to_index": k2i.tolist(), "reads": reads}


if __name__ == "__main__":
    for name, fn in (("tiny_index.json", tiny_index), ("reads_small.json", reads_small)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")
