"""The whole `bucketmap` / `bucketmap_align` tool in plain Python.  TEST INFRASTRUCTURE ONLY.

A third statement of the reference's behaviour, written from the reference's text (file:line below are
relative to /root/reference/bucket_map/) with lists, dicts and Python integers only -- it shares nothing
with oracle/*.c, with oracle/bm_oracle_np.py, with bucket-map_amd/host/*.h or with the kernels.  It is
what tests/golden/make_golden.py runs to produce sam_small.json, the fixture that pins

  * the bucket loop's ordering contract                 locator/bucket_locator.h:651-693
  * _filter_best_locations                              locator/bucket_locator.h:350-405
  * the .bucket_id -> @SQ collapse and bucket offsets   locator/bucket_locator.h:473-503
  * the SAM record fields of both binaries              locator/bucket_locator.h:544-600

independently of the C++ host code that ships (the oracle-backed tool compiles that same host code, so
"GPU tool == oracle-backed tool" compares those four with themselves).

Slow on purpose: everything is a loop.  Only for genomes of a few thousand bases.
"""
import math
import struct

RANK = {}
for _chars, _r in (("AaRrWwMmDdHhVvNn", 0), ("CcYySsBb", 1), ("GgKk", 2), ("TtUu", 3)):
    for _c in _chars:
        RANK[_c] = _r


def rank(c):
    """SeqAn3 dna4 assign_char: anything that is not C/G/T-like folds to A."""
    return RANK.get(c, 0)


def f32(x):
    """round a Python float to float32"""
    return struct.unpack("f", struct.pack("f", x))[0]


def i32(x):
    """wrap to a 32-bit signed int (the reference mixes unsigned and int, then stores into int keys)"""
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x


def sampler(n, ub):
    """Sampler::sample_deterministically, utils.h:160-178 (ub == 0 is never used by the fixtures)."""
    assert ub > 0 or n == 0
    delta = 0.0 if n == 1 else float(ub + 1) / (n - 1)
    return [int(math.floor(i * delta)) for i in range(n - 1)] + [ub]


def kmer_hashes(seq, k):
    """views::kmer_hash(ungapped{k}): big-endian base 4"""
    out = []
    for j in range(len(seq) - k + 1):
        h = 0
        for c in seq[j:j + k]:
            h = h * 4 + rank(c)
        out.append(h)
    return out


def revcomp_hash(h, k):
    """utils.h:291-302"""
    digits = [(h >> (2 * (k - 1 - i))) & 3 for i in range(k)]
    out = 0
    for d in reversed(digits):
        out = out * 4 + (3 - d)
    return out


def cut_buckets(records, bucket_len, read_len):
    """iterate_through_buckets, utils.h:72-97: (record index, start, end) of every kept bucket."""
    out = []
    for r, (_, seq) in enumerate(records):
        total = f32(float(len(seq)))
        n = int(math.ceil(f32(total / f32(float(bucket_len)))))
        for i in range(n):
            start = i * bucket_len
            end = min(start + bucket_len + read_len, len(seq))
            if end - start <= read_len:
                continue
            out.append((r, start, end))
    return out


def awk_bucket_num(records, bucket_len):
    """CMakeLists.txt:15-46: sum of ceil(len / bucket_len) over the records, in double arithmetic."""
    return sum(int(math.ceil(len(seq) / bucket_len)) for _, seq in records if len(seq))


class Tool:
    """Parameters as main.cpp:21-38,202-218 derives them; `align` = the BM_ALIGN build."""

    def __init__(self, records, *, bucket_len, read_len, q, k, S, e, d, b, n, p, u, align=False):
        self.records, self.bucket_len, self.read_len = records, bucket_len, read_len
        self.q, self.k, self.S, self.p, self.u, self.align = q, k, S, p, u, align
        self.F = int(math.ceil(f32(f32(e) * f32(float(S)))))                     # main.cpp:207
        self.min_q = b * k                                                      # q_gram_mapper.h:303, bucket_locator.h:431
        self.allowed_mismatch = int(math.ceil(f32(f32(e) * f32(float(p)))))     # bucket_locator.h:419
        self.allowed_indel = int(math.ceil(f32(f32(n) * f32(float(read_len)))))  # :420
        self.n = f32(n)
        self.buckets = cut_buckets(records, bucket_len, read_len)
        self.NB = awk_bucket_num(records, bucket_len)
        self.threshold = int(f32(f32(d) * f32(float(self.NB))))                 # q_gram_mapper.h:163
        # bucket_indexer.h:49-61 with -f 1 (every q-gram kept, row index = hash): row g = set of buckets
        self.rows = [0] * (4 ** q)
        self.bucket_seq = []
        for bi, (r, s, e_) in enumerate(self.buckets):
            seq = records[r][1][s:e_]
            self.bucket_seq.append(seq)
            for g in kmer_hashes(seq, q):
                self.rows[g] |= 1 << bi
        # distinguishability_filter::read, q_gram_mapper.h:171-186
        self.zeros = [self.NB - bin(row).count("1") for row in self.rows]

    # ---- mapper -------------------------------------------------------------------------------------
    def _query(self, hashes):
        """q_gram_mapper::query (:380-412) + fault_tolerate_filter (:69-102), as integer miss counts."""
        misses = [0] * self.NB
        qmask = 4 ** self.q - 1
        for h in hashes:
            hit = (1 << self.NB) - 1
            for i in range(self.k - self.q + 1):
                hit &= self.rows[(h >> (2 * i)) & qmask]
            for bkt in range(self.NB):
                if not (hit >> bkt) & 1:
                    misses[bkt] += 1
        best = min(misses)
        if best >= self.F:
            return []
        return [bkt for bkt in range(self.NB) if misses[bkt] == best]

    def query_sequence(self, seq, qual):
        """q_gram_mapper::query_sequence (:414-480)"""
        qmask = 4 ** self.q - 1
        good = []
        for j, h in enumerate(kmer_hashes(seq, self.k)):
            dist = any(self.zeros[(h >> (2 * i)) & qmask] >= self.threshold for i in range(self.k - self.q + 1))
            if dist and sum(ord(c) - 33 for c in qual[j:j + self.k]) >= self.min_q:
                good.append(h)
        if len(good) < 0.2 * self.S:
            return [], []
        smp = [good[x] for x in sampler(self.S, len(good) - 1)]
        fwd = self._query(smp)
        rc = self._query([revcomp_hash(h, self.k) for h in smp])
        return ([] if len(fwd) > 30 else fwd), ([] if len(rc) > 30 else rc)

    def window_starts(self, length):
        """q_gram_mapper.h:510-516 / bucket_locator.h:303-308"""
        if length > 2 * self.read_len:
            return sampler(5, length - self.read_len - 1)
        return [0]

    def map(self, reads):
        """q_gram_mapper::map (:483-557): per bucket, (read, window start) in (read, window) order."""
        orig = [[] for _ in range(self.NB)]
        rc = [[] for _ in range(self.NB)]
        for ri, (_, seq, qual) in enumerate(reads):
            for st in self.window_starts(len(seq)):
                end = min(st + self.read_len, len(seq))
                f, r = self.query_sequence(seq[st:end], qual[st:end])
                for bkt in f:
                    orig[bkt].append((ri, st))
                for bkt in r:
                    rc[bkt].append((ri, st))
        return orig, rc

    # ---- locator ------------------------------------------------------------------------------------
    def prepare(self, reads):
        """_prepare_read_query (:292-347): (read, window start) -> (segment length, positions, hashes)"""
        rec = {}
        for ri, (_, seq, qual) in enumerate(reads):
            for st in self.window_starts(len(seq)):
                end = min(st + self.read_len, len(seq))
                s, ql = seq[st:end], qual[st:end]
                hs = kmer_hashes(s, self.k)
                if not hs:
                    continue
                good = [j for j in range(len(hs)) if sum(ord(c) - 33 for c in ql[j:j + self.k]) >= self.min_q]
                if not good:
                    good = list(range(len(hs)))
                pos = [good[x] for x in sampler(self.p, len(good) - 1)]
                rec[(ri, st)] = (len(s), pos, [hs[j] for j in pos])
        return rec

    def find_offset(self, occurrences, record, reverse_complement):
        """_find_offset (:209-290).  `occurrences`: hash -> offsets in the order equal_range yields them."""
        length, index, hashes = record
        votes = {}                                              # std::map<int, unsigned>
        for i in range(self.p):
            si = self.p - 1 - i if reverse_complement else i
            kmer, at = hashes[si], index[si]
            if reverse_complement:
                kmer, at = revcomp_hash(kmer, self.k), length - self.k - at
            occ = occurrences.get(kmer, [])
            if not votes:
                for o in occ:
                    pos = i32(o - at)
                    votes[pos] = votes.get(pos, 0) + 1
            else:
                for o in occ:
                    pos = i32(o - at)
                    near = [key for key in sorted(votes) if pos - self.allowed_indel <= key <= pos + self.allowed_indel]
                    for key in near:
                        votes[key] += 1
                    if not near:
                        votes[pos] = votes.get(pos, 0) + 1
        if votes:
            top = max(votes.values())
            key = min(kk for kk, v in votes.items() if v == top)
            if top >= self.p - self.allowed_mismatch and key >= 0:
                return key, top
        return -1, 0

    def locate_reads(self, reads):
        """_locate (:613-705): per read, the locations in the order the bucket loop appends them."""
        orig, rc = self.map(reads)
        rec = self.prepare(reads)
        res = [[] for _ in reads]
        for bi in range(len(orig)):
            if not orig[bi] and not rc[bi]:
                continue
            if bi >= len(self.bucket_seq):
                continue                                        # a padding bucket id holds no sequence
            # _create_kmer_index (:162-177): libstdc++'s unordered_multimap yields equal keys in DESCENDING offset
            occ = {}
            for off, h in enumerate(kmer_hashes(self.bucket_seq[bi], self.k)):
                occ.setdefault(h, []).insert(0, off)
            for seg in orig[bi]:
                if seg not in rec:
                    continue
                off, v = self.find_offset(occ, rec[seg], False)
                if off > 0:                                     # :674
                    res[seg[0]].append((bi, off - seg[1], seg[1], v, True))
            for seg in reversed(rc[bi]):                        # :682
                if seg not in rec:
                    continue
                off, v = self.find_offset(occ, rec[seg], True)
                if off > 0:                                     # :686
                    seg_off = len(reads[seg[0]][1]) - seg[1] - rec[seg][0]
                    res[seg[0]].append((bi, off - seg_off, seg[1], v, False))
        return res

    def filter_best_locations(self, locs, read_len):
        """_filter_best_locations (:350-405)"""
        votes = {}                                              # std::map<(bucket, offset, is_orig), votes>
        for bucket, off, _, v, is_orig in locs:
            if not votes:
                votes[(bucket, off, is_orig)] = v
                continue
            span = f32(f32(float(read_len)) * self.n)
            lo = int(f32(f32(float(off)) - span))               # int = float, truncated towards zero (:365-366)
            hi = int(f32(f32(float(off)) + span))
            found = False
            for key in sorted(votes):
                if key[0] == bucket and lo <= key[1] <= hi and key[2] == is_orig:
                    votes[key] += v
                    found = True
            if not found:
                votes[(bucket, off, is_orig)] = v               # assignment, not accumulation (:380)
        if not votes:
            return []
        top = max(votes.values())
        return [(key[0], key[1], 0, votes[key], key[2]) for key in sorted(votes) if votes[key] == top]

    def sam_header(self):
        """.bucket_id -> @SQ lines and per-bucket offsets (:473-503)"""
        names, offsets, refs = [], [], []
        last, idx = "", 0
        for r, _, _ in self.buckets:
            name = self.records[r][0].split(" ")[0]
            if name != last:
                if idx:
                    refs.append((last, idx * self.bucket_len))
                last, idx = name, 0
            names.append(name)
            offsets.append(idx * self.bucket_len)
            idx += 1
        if idx:
            refs.append((last, idx * self.bucket_len))
        return names, offsets, refs

    def run(self, reads):
        """locate (:455-611): (@SQ list, records); a record = (QNAME, FLAG, RNAME, POS, MAPQ, CIGAR, SEQ, QUAL)."""
        names, offsets, refs = self.sam_header()
        out = []
        located = self.locate_reads(reads)
        for ri, (rid, seq, qual) in enumerate(reads):
            folded = "".join("ACGT"[rank(c)] for c in seq)       # record.sequence() is a dna4 vector
            if self.align:
                for bucket, off, _, _, is_orig in located[ri]:  # no _filter_best_locations here (:538-541)
                    text = self.bucket_seq[bucket]
                    start = min(max(off, 0), len(text))          # negative offsets: see DESIGN.md (clipped)
                    width = min(len(seq) + 1 + int(f32(self.n * f32(float(len(seq))))), len(text) - start)
                    score, begin, cigar = align(text[start:start + width], seq, not is_orig)
                    mapq = (60 + score) & 0xFFFFFFFF                                      # size_t = 60u + int (:570)
                    if mapq < self.u:
                        continue
                    pos = begin + offsets[bucket] + max(off, 0)                           # :576
                    out.append((rid, 0 if is_orig else 16, names[bucket], pos + 1, mapq & 0xFF, cigar or "*", folded, qual))
            else:
                for bucket, off, _, v, is_orig in self.filter_best_locations(located[ri], len(seq)):
                    out.append((rid, 0 if is_orig else 16, names[bucket], offsets[bucket] + off + 1, min(60, 6 * v),
                                "*", folded, qual))
        return refs, out


def align(text, query, text_rc):
    """bucket_locator.h:520-528,562-576: edit distance of the whole query against the best substring of the
    text; ties as include/bmv.h states them (last minimal end column; diagonal, up, left)."""
    t = [rank(c) for c in text]
    if text_rc:
        t = [3 - r for r in reversed(t)]
    q = [rank(c) for c in query]
    n, m = len(t), len(q)
    H = [[0] * (n + 1) for _ in range(m + 1)]
    for i in range(1, m + 1):
        H[i][0] = i
        for j in range(1, n + 1):
            H[i][j] = min(H[i - 1][j - 1] + (q[i - 1] != t[j - 1]), H[i - 1][j] + 1, H[i][j - 1] + 1)
    j = max(range(n + 1), key=lambda c: (-H[m][c], c))
    score, i, ops = -H[m][j], m, []
    while i > 0:
        if j > 0 and H[i][j] == H[i - 1][j - 1] + (q[i - 1] != t[j - 1]):
            ops.append("M"); i -= 1; j -= 1
        elif H[i][j] == H[i - 1][j] + 1:
            ops.append("I"); i -= 1
        else:
            ops.append("D"); j -= 1
    ops.reverse()
    cigar, a = "", 0
    while a < len(ops):
        e = a
        while e < len(ops) and ops[e] == ops[a]:
            e += 1
        cigar += f"{e - a}{ops[a]}"
        a = e
    return score, j, cigar
