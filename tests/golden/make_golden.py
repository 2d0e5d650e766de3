"""Generates the committed fixtures tests/golden/*.json.

The reference ships no golden vectors for this path and cannot be run here, so these vectors are
produced by the NUMPY formulation of the algorithm (oracle/bm_oracle_np.py: integer miss counts +
argmin), with the index built by a brute-force numpy indexer written here.  They pin the C oracle
(which applies the reference's bit-plane update literally) and, through it, the HIP path.

align_small.json and sampling_small.json pin the two "next" paths (alignment verification, the locator's
k-mer sampling) the same way: expected values from plain-Python restatements written here -- lists and
loops, sharing nothing with oracle/*.c or the kernels.

    python tests/golden/make_golden.py        # rewrites the four fixtures
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


RANK = {c: r for r, cs in enumerate(("AaRrWwMmDdHhVvNn", "CcYySsBb", "GgKk", "TtUu")) for c in cs}


def py_align(text, query, text_rc):
    """bucket_locator.h:520-528,562-576 in plain Python: semi-global edit distance of the whole query against
    the best substring of the text; ties as include/bmv.h states them (last minimal end column; diagonal, up,
    left).  Returns (score, begin, cigar)."""
    t = [RANK.get(c, 0) for c in text]
    if text_rc:
        t = [3 - r for r in reversed(t)]
    q = [RANK.get(c, 0) for c in query]
    n, m = len(t), len(q)
    H = [[0] * (n + 1) for _ in range(m + 1)]
    for i in range(1, m + 1):
        H[i][0] = i
        for j in range(1, n + 1):
            H[i][j] = min(H[i - 1][j - 1] + (q[i - 1] != t[j - 1]), H[i - 1][j] + 1, H[i][j - 1] + 1)
    j = max(range(n + 1), key=lambda c: (-H[m][c], c))          # smallest score, then the largest column
    score, i, ops = -H[m][j], m, []
    while i > 0:
        if j > 0 and H[i][j] == H[i - 1][j - 1] + (q[i - 1] != t[j - 1]):
            ops.append("M"); i -= 1; j -= 1
        elif H[i][j] == H[i - 1][j] + 1:
            ops.append("I"); i -= 1
        else:
            ops.append("D"); j -= 1
    ops.reverse()
    cigar, k = "", 0
    while k < len(ops):
        e = k
        while e < len(ops) and ops[e] == ops[k]:
            e += 1
        cigar += f"{e - k}{ops[k]}"
        k = e
    return score, j, cigar


def align_small():
    rng = np.random.default_rng(20240012)
    alphabet = "ACGT"
    cases = []
    for c in range(260):
        n, m = int(rng.integers(0, 70)), int(rng.integers(0, 40))
        text = "".join(alphabet[i] for i in rng.integers(0, 4, n))
        if c % 3 == 0 and n > m + 4 and m > 3:                  # a query cut from the text and damaged
            at = int(rng.integers(0, n - m))
            src = text[at:at + m]
            if c % 2:
                src = "".join("TGCA"["ACGT".index(x)] for x in reversed(src))
            query = "".join(x if rng.random() > 0.1 else alphabet[int(rng.integers(0, 4))] for x in src)
            if m > 8:
                cut = int(rng.integers(1, m - 1))
                query = query[:cut] + query[cut + 1:] if c % 4 else query[:cut] + "G" + query[cut:]
            rc = bool(c % 2)
        else:
            query = "".join("ACGTNacgtRY"[i] for i in rng.integers(0, 11, m))
            rc = bool(rng.integers(0, 2))
        score, begin, cigar = py_align(text, query, rc)
        cases.append({"text": text, "query": query, "rc": rc, "score": score, "begin": begin, "cigar": cigar})
    return {"cases": cases}


def py_sample(bases, quals, k, p, minq):
    """bucket_locator.h:317-343 in plain Python: (has, positions, hashes) of one window."""
    nk = len(bases) - k + 1 if len(bases) >= k else 0
    if nk == 0:
        return 0, [0] * p, [0] * p
    good = [j for j in range(nk) if sum(ord(c) - 33 for c in quals[j:j + k]) >= minq]
    if not good:
        good = list(range(nk))
    ub = len(good) - 1
    delta = (ub + 1) / (p - 1) if p != 1 else 0.0
    at = [int(np.floor(s * delta)) for s in range(p - 1)] + [ub]
    pos = [good[a] for a in at]
    hashes = []
    for j in pos:
        h = 0
        for c in bases[j:j + k]:
            h = (h << 2) | RANK.get(c, 0)
        hashes.append(h)
    return 1, pos, hashes


def sampling_small():
    rng = np.random.default_rng(20240013)
    groups = []
    for k, p, minq in ((12, 10, 25 * 12), (9, 5, 0), (16, 20, 10 * 16), (12, 10, 10 ** 6), (5, 1, 60), (3, 64, 20)):
        windows = []
        for w in range(40):
            n = int(rng.integers(0, 90)) if w % 4 else [0, k - 1, k, k + 1, 150][w // 4 % 5]
            bases = "".join("ACGTNacgt"[i] for i in rng.integers(0, 9, n))
            quals = "".join(chr(int(x)) for x in rng.integers(33, 75, n))
            if w % 7 == 3:
                quals = "I" * n
            has, pos, hashes = py_sample(bases, quals, k, p, minq)
            windows.append({"bases": bases, "quals": quals, "has": has, "pos": pos, "hash": hashes})
        groups.append({"k": k, "p": p, "min_base_quality": minq, "windows": windows})
    return {"groups": groups}


if __name__ == "__main__":
    for name, fn in (("tiny_index.json", tiny_index), ("reads_small.json", reads_small),
                     ("align_small.json", align_small), ("sampling_small.json", sampling_small)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")
