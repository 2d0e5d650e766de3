"""Generates the committed fixtures tests/golden/*.json.

The reference ships no golden vectors for this path and cannot be run here, so these vectors are
produced by the NUMPY formulation of the algorithm (oracle/bm_oracle_np.py: integer miss counts +
argmin), with the index built by a brute-force numpy indexer written here.  They pin the C oracle
(which applies the reference's bit-plane update literally) and, through it, the HIP path.

align_small.json and sampling_small.json pin the two "next" paths (alignment verification, the locator's
k-mer sampling) the same way: expected values from plain-Python restatements written here -- lists and
loops, sharing nothing with oracle/*.c or the kernels.

    python tests/golden/make_golden.py        # rewrites the five fixtures (or only those named)
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


SAM_SMALL_FLAGS = dict(bucket_len=256, read_len=60, q=5, k=7, S=8, e=0.5, d=0.5, b=20, n=0.05, p=8, u=40)


def sam_small():
    """A tiny genome and 60-odd reads pushed through tests/golden/py_bucketmap.py (the whole tool in plain
    Python): expected @SQ lines and SAM records of `bucketmap` and of `bucketmap_align`.  Built to hit the
    order-sensitive host code: records whose names share the text before the first blank (one @SQ, offsets
    carried across), a record whose tail bucket is dropped (NB > kept buckets), a duplicated stretch (vote ties:
    two records for one read), hits on both strands, hits at bucket offset 0 (dropped, :674/:686) and at the
    overlap of two buckets, 5-window long reads whose proposals merge, N / lower-case bases (SEQ is written
    folded to ACGT), low and noisy qualities, reads shorter than k and reads from nowhere."""
    import py_bucketmap as pb
    rng = np.random.default_rng(20240014)
    alpha = "ACGT"

    def rnd(n):
        return "".join(alpha[i] for i in rng.integers(0, 4, n))

    def rc(s):
        return "".join("TGCA"["ACGT".index(c)] for c in reversed(s))

    chr_b = rnd(1000)
    chr_c = rnd(552)
    chr_c = chr_c[:150] + chr_b[400:530] + chr_c[280:]          # 130 bases of chrB again inside chrC
    records = [("chrA part1 of two", rnd(1500)), ("chrA part2", rnd(700)), ("chrB", chr_b), ("chrC tail dropped", chr_c)]
    reads = []

    def add(name, seq, qual=None, flip=False):
        seq = rc(seq) if flip else seq
        reads.append((name, seq, qual if qual is not None else "E" * len(seq)))

    def damaged(s, subs=0, dele=None, ins=None):
        s = list(s)
        for _ in range(subs):
            at = int(rng.integers(0, len(s)))
            s[at] = alpha[(alpha.index(s[at]) + int(rng.integers(1, 4))) % 4]
        if dele is not None:
            del s[dele]
        if ins is not None:
            s.insert(ins, "G")
        return "".join(s)

    for i in range(14):                                         # plain short reads, both strands, 0-3 substitutions
        r = int(rng.integers(0, 4))
        at = int(rng.integers(1, len(records[r][1]) - 61))
        add(f"plain{i}", damaged(records[r][1][at:at + 60], subs=i % 4), flip=bool(i % 2))
    add("record_start", records[0][1][0:60])                    # offset 0 of bucket 0: never reported
    add("record_start_rc", records[2][1][0:60], flip=True)
    add("bucket_edge", records[0][1][256:316])                  # offset 0 of bucket 1 = offset 256 of bucket 0
    add("bucket_edge_rc", records[0][1][512:572], flip=True)
    add("overlap", records[0][1][280:340])                      # lies in buckets 0 and 1
    add("part2", records[1][1][300:360])                        # RNAME chrA, POS continues after part1's buckets
    add("part2_rc", records[1][1][520:580], flip=True)
    add("dup", chr_b[430:490])                                  # also in chrC: two best locations
    add("dup_rc", chr_b[440:500], flip=True)
    add("dup_partial", chr_b[380:440])                          # only 40 of its 60 bases are in chrC
    add("tail_bucket", records[3][1][470:530])                  # lives in chrC's last kept bucket only
    add("with_N", records[2][1][100:110] + "NNNN" + records[2][1][114:160])
    add("lower_case", records[2][1][700:760].lower())
    add("iupac", records[0][1][900:920] + "RYKM" + records[0][1][924:960])
    add("low_quality", records[0][1][1000:1060], qual="#" * 60)
    add("noisy_quality", records[0][1][1100:1160], qual="".join(chr(int(x)) for x in rng.integers(40, 75, 60)))
    add("half_bad_quality", records[1][1][100:160], qual="#" * 30 + "I" * 30)
    add("deletion", damaged(records[2][1][200:261], dele=30))
    add("insertion_rc", damaged(records[2][1][300:359], ins=25), flip=True)
    add("len61", records[0][1][600:661])
    add("len100_rc", records[0][1][700:800], flip=True)
    add("len120", records[1][1][50:170])
    add("len121", records[2][1][500:621])                       # > 2 * read_len: five windows
    add("long200", damaged(records[0][1][100:300], subs=3))
    add("long300_rc", damaged(records[0][1][900:1200], subs=5, dele=150), flip=True)
    add("long250_two_records", records[1][1][500:700] + records[2][1][0:50])   # chimeric
    add("long400", damaged(records[2][1][350:750], subs=6, ins=200))
    add("short5", "ACGTA")
    add("len8", records[0][1][40:48])                           # two k-mers (one would hit the Sampler's ub == 0 quirk)
    add("len20", records[0][1][40:60])
    for i in range(4):
        add(f"junk{i}", rnd(60 + 40 * i))
    for i in range(10):                                         # noisier reads: 4-8 substitutions, some indels
        r = int(rng.integers(0, 4))
        at = int(rng.integers(1, len(records[r][1]) - 62))
        add(f"noisy{i}", damaged(records[r][1][at:at + 61], subs=4 + i % 5, dele=20 if i % 3 == 0 else None),
            flip=bool(i % 2))
    out = {"flags": SAM_SMALL_FLAGS, "records": [list(r) for r in records], "reads": [list(r) for r in reads]}
    for key, align in (("bucketmap", False), ("bucketmap_align", True)):
        tool = pb.Tool(records, align=align, **SAM_SMALL_FLAGS)
        refs, sam = tool.run(reads)
        out[key] = {"num_buckets": tool.NB, "kept_buckets": len(tool.buckets), "sq": [list(r) for r in refs],
                    "sam": [list(r) for r in sam]}
    return out


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    for name, fn in (("tiny_index.json", tiny_index), ("reads_small.json", reads_small),
                     ("align_small.json", align_small), ("sampling_small.json", sampling_small),
                     ("sam_small.json", sam_small)):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")
