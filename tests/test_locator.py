"""The locator candidate scan: CPU oracle sanity (no GPU) and GPU == oracle parity (gpu).

Bar for the GPU tests: bit-exact (offset, votes) for every candidate -- the vote of _find_offset is
order dependent (bucket_locator.h:233-274), so repeats are the interesting inputs.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle import bm_oracle_np as onp
from oracle import oracle_c as oc

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LETTERS = np.frombuffer(b"ACGT", np.uint8)


def revcomp(seq_codes):
    return 3 - seq_codes[::-1]


def make_case(rng, *, n_buckets, bucket_len, read_len, n_reads, k=12, p=10, motif=None, sub=0.01, decoys=True, qual_b=25):
    """Genome of n_buckets buckets (optionally built from a short repeated motif), reads with their sampled
    locator k-mers (bucket_locator.h:292-347 restated in numpy) and candidate (window, bucket, strand)
    pairs: the true one plus decoys."""
    size = bucket_len + read_len
    if motif is None:
        genome = rng.integers(0, 4, n_buckets * bucket_len + read_len).astype(np.uint8)
    else:
        unit = rng.integers(0, 4, motif).astype(np.uint8)
        genome = np.tile(unit, (n_buckets * bucket_len + read_len) // motif + 1)[: n_buckets * bucket_len + read_len].copy()
        flips = rng.random(len(genome)) < 0.02                     # a few point differences between copies
        genome[flips] = rng.integers(0, 4, flips.sum())
    bstart = (np.arange(n_buckets) * bucket_len).astype(np.uint64)
    blen = np.full(n_buckets, size, np.uint32)
    blen[-1] = len(genome) - int(bstart[-1])
    sh, sp, sl, pb, pw, pr, truth = [], [], [], [], [], [], []
    for r in range(n_reads):
        b = int(rng.integers(0, n_buckets))
        start = int(rng.integers(1, int(blen[b]) - read_len - 1))
        seq = genome[int(bstart[b]) + start: int(bstart[b]) + start + read_len].copy()
        for _ in range(int(rng.poisson(sub * read_len))):
            seq[int(rng.integers(0, read_len))] = rng.integers(0, 4)
        rc = bool(rng.integers(0, 2))
        if rc:
            seq = revcomp(seq)
        text = LETTERS[seq]
        quals = np.full(read_len, ord("E"), np.uint8)
        if r % 4 == 1:
            quals = rng.integers(33 + 10, 33 + 41, read_len).astype(np.uint8)
        hs, qs = onp.kmer_hashes(text, k), onp.kmer_qualities(quals, k)
        good = np.nonzero(qs >= qual_b * k)[0]
        if len(good) == 0:
            good = np.arange(len(hs))
        pos = [int(good[i]) for i in onp.sample_positions(p, len(good) - 1)]
        sh.append([int(hs[j]) for j in pos]); sp.append(pos); sl.append(read_len)
        cands = [(b, rc)]
        if decoys:
            cands += [(b, not rc), (int(rng.integers(0, n_buckets)), rc)]
        for cb, crc in cands:
            pb.append(cb); pw.append(r); pr.append(int(crc))
        truth.append((b, start, rc))
    order = np.argsort(np.array(pb), kind="stable")                # candidates grouped by bucket
    pb, pw, pr = np.array(pb, np.uint32)[order], np.array(pw, np.uint32)[order], np.array(pr, np.uint8)[order]
    return dict(genome=LETTERS[genome], bstart=bstart, blen=blen, sh=np.array(sh, np.uint32), sp=np.array(sp, np.uint16),
                sl=np.array(sl, np.uint32), pb=pb, pw=pw, pr=pr, truth=truth, k=k, p=p)


def oracle(case, mismatch=4, indel=6):
    return oc.locate(case["k"], case["p"], mismatch, indel, case["genome"], case["bstart"], case["blen"], case["sh"],
                     case["sp"], case["sl"], case["pb"], case["pw"], case["pr"])


# ------------------------------------------------------------------------------------------ CPU

def test_unordered_multimap_order_assumption():
    # the one assumption the oracle imports from libstdc++: equal keys come back in descending offset
    exe = os.path.join(ROOT, "tests", "cpp", "umm_order")
    subprocess.run(["make", "-C", ROOT, "tests/cpp/umm_order"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "OK"


def test_oracle_finds_true_offsets():
    rng = np.random.default_rng(1)
    case = make_case(rng, n_buckets=6, bucket_len=4096, read_len=150, n_reads=120)
    off, votes = oracle(case)
    hit = 0
    for i in range(len(case["pb"])):
        b, start, rc = case["truth"][int(case["pw"][i])]
        if int(case["pb"][i]) == b and bool(case["pr"][i]) == rc:
            hit += int(abs(int(off[i]) - start) <= 6 and votes[i] >= 6)
        else:
            assert off[i] == -1 and votes[i] == 0               # decoys: wrong strand / wrong bucket
    assert hit >= 0.95 * len(case["truth"])


def test_oracle_hand_traced_vote():
    # bucket = ACGT x 8 (32 bases), k=4, p=2.  Window "ACGTACGT" sampled at positions 0 and 4 (both "ACGT" = 27).
    # Occurrences of ACGT: offsets 0,4,...,28, visited DESCENDING.  Sample 0 (idx 0): proposals 28,24,...,0 with 1
    # vote each.  Sample 1 (idx 4): occurrence o proposes o-4; with indel 0 it votes for the existing key o-4
    # (o = 28..4) and creates -4 for o = 0.  Keys 0..24 have 2 votes, 28 and -4 have 1: max 2, smallest key 0.
    genome = np.frombuffer(b"ACGT" * 8, np.uint8)
    off, votes = oc.locate(4, 2, 0, 0, genome, [0], [32], [[27, 27]], [[0, 4]], [8], [0], [0], [0])
    assert (int(off[0]), int(votes[0])) == (0, 2)
    # with indel 4 every sample-1 occurrence (positions P = 24, 20, ..., 0, -4) votes for ALL proposals within
    # +-4: key K gets one vote per P with |P-K| <= 4.  K = 0 gets P in {4, 0, -4} (so P = -4 creates no key),
    # K = 4..20 get three, K = 24 gets {24, 20}, K = 28 gets {24}.  Max 1 + 3 = 4 votes, smallest such key = 0.
    off, votes = oc.locate(4, 2, 0, 4, genome, [0], [32], [[27, 27]], [[0, 4]], [8], [0], [0], [0])
    assert (int(off[0]), int(votes[0])) == (0, 4)
    # order dependence: sampled at positions 4 then 0 instead, the first sample proposes -4, 0, ..., 24; the
    # second (P = 28, ..., 0) then creates nothing new either, but now key -4 exists and collects {0}: the
    # winner must still be a key >= 0 with the most votes
    off, votes = oc.locate(4, 2, 0, 4, genome, [0], [32], [[27, 27]], [[4, 0]], [8], [0], [0], [0])
    assert int(votes[0]) == 4 and int(off[0]) >= 0
    # reverse-complement candidate of the same window: ACGT is its own reverse complement, positions mirror
    off, votes = oc.locate(4, 2, 0, 0, genome, [0], [32], [[27, 27]], [[0, 4]], [8], [0], [0], [1])
    assert (int(off[0]), int(votes[0])) == (0, 2)
    # too few votes: p - allowed_mismatch = 2 needed, a window that matches once only gets 1
    off, votes = oc.locate(4, 2, 0, 0, genome, [0], [32], [[27, 0]], [[0, 4]], [8], [0], [0], [0])
    assert (int(off[0]), int(votes[0])) == (-1, 0)


# ------------------------------------------------------------------------------------------ GPU

def gpu_scan(case, mismatch=4, indel=6):
    from bucket_map_amd import locate
    s = locate.LocatorScan(case["k"], case["p"], mismatch, indel, int(case["blen"].max()))
    s.load_genome(case["genome"], case["bstart"], case["blen"])
    off, votes = s.locate(case["sh"], case["sp"], case["sl"], case["pb"], case["pw"], case["pr"])
    st = s.stats()
    s.close()
    return off, votes, st


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [
    ("random genome", dict(n_buckets=40, bucket_len=8192, read_len=150, n_reads=1500)),
    ("reference geometry", dict(n_buckets=6, bucket_len=65536, read_len=300, n_reads=600)),
    ("tandem repeats", dict(n_buckets=4, bucket_len=8192, read_len=150, n_reads=300, motif=97)),
    ("short motif: thousands of occurrences", dict(n_buckets=2, bucket_len=16384, read_len=100, n_reads=60, motif=23)),
    ("one hot bucket: chunking", dict(n_buckets=1, bucket_len=32768, read_len=150, n_reads=1200, decoys=False)),
    ("k=16 p=20", dict(n_buckets=8, bucket_len=4096, read_len=200, n_reads=300, k=16, p=20)),
    ("k=9 p=5", dict(n_buckets=8, bucket_len=4096, read_len=120, n_reads=300, k=9, p=5)),
])
def test_gpu_scan_equals_oracle(name, kw):
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    case = make_case(rng, **kw)
    p = case["p"]
    mismatch, indel = int(np.ceil(0.4 * p)), 6
    o_ref, v_ref = oracle(case, mismatch, indel)
    o_got, v_got, st = gpu_scan(case, mismatch, indel)
    bad = np.nonzero((o_ref != o_got) | (v_ref != v_got))[0]
    assert bad.size == 0, f"{name}: {bad.size} candidates differ, first {bad[:5]}: ref {o_ref[bad[:5]]}/{v_ref[bad[:5]]} got {o_got[bad[:5]]}/{v_got[bad[:5]]}"
    assert st["occurrences"] >= (o_ref >= 0).sum()
    if "motif" in kw:
        assert st["occurrences"] > 20 * len(case["pb"])          # repeats really produce many occurrences
        assert st["heavy_candidates"] > 0.3 * len(case["pb"])    # ... and their candidates take the dense-bitmap kernel
    else:
        assert st["heavy_candidates"] < 0.05 * len(case["pb"])   # one thread each (the light kernel's room grows with p)


@pytest.mark.gpu
@pytest.mark.parametrize("budget", [500, 20_000, 400_000])
def test_gpu_scan_in_groups_within_an_occurrence_budget(budget, monkeypatch):
    """When the occurrences of a call do not fit the budget (BML_MAX_OCC; 2^31 by default -- 10 M reads on the genome-like
    genome bring 11 G), the candidates are cut into groups of whole chunks that are scanned and replayed one after the other:
    same offsets and votes, whatever the budget -- even one smaller than a single chunk's occurrences."""
    rng = np.random.default_rng(4242)
    case = make_case(rng, n_buckets=6, bucket_len=8192, read_len=150, n_reads=400, motif=97)
    o_ref, v_ref = oracle(case)
    monkeypatch.setenv("BML_MAX_OCC", str(budget))
    o_got, v_got, st = gpu_scan(case)
    assert np.array_equal(o_ref, o_got) and np.array_equal(v_ref, v_got)
    assert st["occurrences"] > 4 * budget or budget > 100_000    # the small budgets really force groups
    assert st["heavy_candidates"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("first", [1, 5000])
def test_gpu_scan_again_when_the_first_buffer_was_too_small(first, monkeypatch):
    """The first scan runs with room for two occurrences per sample; when repeats bring more it has still counted every
    candidate's and placed their segments, and the second scan only writes them (BML_FIRST_OCC: the first buffer's size)."""
    rng = np.random.default_rng(777)
    case = make_case(rng, n_buckets=5, bucket_len=8192, read_len=150, n_reads=300, motif=53)
    o_ref, v_ref = oracle(case)
    monkeypatch.setenv("BML_FIRST_OCC", str(first))
    o_got, v_got, st = gpu_scan(case)
    assert st["occurrences"] > first
    assert np.array_equal(o_ref, o_got) and np.array_equal(v_ref, v_got)
    assert st["heavy_candidates"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("bucket_len,read_len,indel", [(262144, 300, 30), (300000, 40000, 700)])
def test_gpu_scan_heavy_candidates_at_long_read_geometry(bucket_len, read_len, indel):
    """BASELINE configs[4]'s bucket length with satellite-like content: a 171-base monomer fills the bucket, every sampled
    k-mer occurs ~1 500 times, a candidate has tens of thousands of occurrences.  (One thread per candidate and a sorted
    array took minutes here.)  Second case: windows so long that the three start-position bitmaps no longer fit LDS and live
    in the workgroup's global scratch."""
    rng = np.random.default_rng(bucket_len)
    case = make_case(rng, n_buckets=2, bucket_len=bucket_len, read_len=read_len, n_reads=12, p=20, motif=171, sub=0.02)
    mismatch = 18                                                # -e 0.9 x -p 20 (benchmark/long_read/benchmark_map.sh:25)
    o_ref, v_ref = oracle(case, mismatch, indel)
    o_got, v_got, st = gpu_scan(case, mismatch, indel)
    assert np.array_equal(o_ref, o_got) and np.array_equal(v_ref, v_got)
    assert st["heavy_candidates"] > 0.5 * len(case["pb"]) and st["occurrences"] > 5000 * len(case["pb"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("BM_SWEEP_SEEDS", "12"))))   # soak: BM_SWEEP_SEEDS=500
def test_gpu_scan_random_sweep(seed):
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(6, 17))
    p = int(rng.choice([1, 2, 5, 10, 20, 33, 64]))
    read_len = int(rng.choice([max(k + 3, 40), 100, 150]))
    kw = dict(n_buckets=int(rng.integers(1, 9)), bucket_len=int(rng.choice([512, 2048, 8192])), read_len=read_len,
              n_reads=int(rng.integers(20, 200)), k=k, p=p, sub=float(rng.choice([0.0, 0.01, 0.05])),
              motif=(None if rng.random() < 0.5 else int(rng.integers(5, 200))), qual_b=int(rng.choice([0, 25, 40])))
    case = make_case(rng, **kw)
    mismatch = int(rng.integers(0, p + 2))
    indel = int(rng.choice([0, 1, 6, 30]))
    o_ref, v_ref = oracle(case, mismatch, indel)
    o_got, v_got, _ = gpu_scan(case, mismatch, indel)
    bad = np.nonzero((o_ref != o_got) | (v_ref != v_got))[0]
    assert bad.size == 0, f"{kw} mismatch={mismatch} indel={indel}: {bad.size} differ, first {bad[:5]}"


@pytest.mark.gpu
def test_gpu_scan_hand_traced():
    from bucket_map_amd import locate
    genome = np.frombuffer(b"ACGT" * 8, np.uint8)
    for indel, want in ((0, (0, 2)), (4, (0, 4))):
        s = locate.LocatorScan(4, 2, 0, indel, 32)
        s.load_genome(genome, [0], [32])
        off, votes = s.locate([[27, 27]], [[0, 4]], [8], [0], [0], [0])
        assert (int(off[0]), int(votes[0])) == want
        s.close()


@pytest.mark.gpu
def test_gpu_scan_errors():
    from bucket_map_amd import locate
    s = locate.LocatorScan(12, 10, 4, 6, 1000)
    with pytest.raises(locate.BmlError):                           # no genome yet
        s.locate(np.zeros((1, 10)), np.zeros((1, 10)), [100], [0], [0], [0])
    g = np.frombuffer(b"ACGT" * 100, np.uint8)
    with pytest.raises(locate.BmlError):                           # bucket longer than max_bucket_bases
        s.load_genome(np.tile(g, 4), [0], [1600])
    s.load_genome(g, [0], [400])
    with pytest.raises(locate.BmlError):                           # unknown bucket
        s.locate(np.zeros((1, 10)), np.zeros((1, 10)), [100], [3], [0], [0])
    off, votes = s.locate(np.zeros((1, 10)), np.zeros((1, 10)), [100], np.zeros(0), np.zeros(0), np.zeros(0))
    assert len(off) == 0
    s.close()


# ------------------------------------------------------------------ _prepare_read_query's sampling (:292-347)

def _sample_case(rng, n, max_len, k):
    lens = rng.integers(0, max_len + 1, n)
    lens[: n // 3] = max_len
    lens[n // 3: n // 3 + 5] = [0, 1, k - 1, k, k + 1][: 5]
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    bases = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.integers(0, 9, int(off[-1]))]
    quals = rng.integers(33, 33 + 42, int(off[-1])).astype(np.uint8)
    # windows inside the reads too (long-read style): every fifth window is a shifted, shortened view
    ws, wl = off[:-1].copy(), lens.astype(np.uint32)
    for w in range(0, n, 5):
        if wl[w] > 20:
            ws[w] += 7
            wl[w] -= 11
    return bases, quals, ws, wl


def test_oracle_sampling_hand_worked():
    # k = 3, p = 4, threshold 3*35 = 105 on phred+33 qualities: window ACGTACGT, qualities I(40) x4 then #(2) x4
    bases = np.frombuffer(b"ACGTACGT", np.uint8)
    quals = np.frombuffer(b"IIII####", np.uint8)
    h, pos, has = oc.sample_windows(3, 4, 105, bases, quals, [0], [8])
    # quality sums of the 6 k-mers: 120 120 82 44 6 6 -> good = {0, 1}; Sampler(4) over 2: floor(i * 2/3) = 0 0 1, last = 1
    assert has.tolist() == [1] and pos[0].tolist() == [0, 0, 1, 1]
    assert h[0].tolist() == [0b000110, 0b000110, 0b011011, 0b011011]          # ACG, ACG, CGT, CGT
    # nothing reaches the threshold: all 6 k-mers are candidates (:330-332): floor(i * 6/3) = 0 2 4, last = 5
    h, pos, has = oc.sample_windows(3, 4, 999, bases, quals, [0], [8])
    assert pos[0].tolist() == [0, 2, 4, 5]
    # shorter than k: no samples
    h, pos, has = oc.sample_windows(3, 4, 0, bases, quals, [0, 6], [2, 2])
    assert has.tolist() == [0, 0] and not h.any() and not pos.any()


@pytest.mark.gpu
@pytest.mark.parametrize("k,p,max_len,minq", [(12, 10, 300, 25 * 12), (12, 20, 300, 10 * 12), (9, 5, 150, 0), (16, 10, 997, 30 * 16),
                                              (12, 10, 300, 10 ** 6), (12, 64, 80, 20 * 12), (12, 10, 16384, 25 * 12)])
def test_gpu_sampling_equals_oracle(k, p, max_len, minq):
    from bucket_map_amd import locate
    rng = np.random.default_rng(k * 1000 + p)
    bases, quals, ws, wl = _sample_case(rng, 40 if max_len > 5000 else 700, max_len, k)
    want = oc.sample_windows(k, p, minq, bases, quals, ws, wl)
    scan = locate.LocatorScan(k, p, 4, 6, 70000)
    got = scan.sample_windows(bases, quals, ws, wl, minq)
    for a, b, what in zip(want, got, ("hash", "position", "has-samples")):
        assert np.array_equal(a, b), what
    scan.close()


def _single_good_kmer_batch(k, b):
    """Windows whose qualities leave EXACTLY ONE k-mer at b*k or above (bucket_locator.h:325-333 then calls
    Sampler::sample_deterministically(0), which returns without sampling, utils.h:165), and windows of exactly k bases
    (one k-mer, whatever its quality).  Returns bases, quals, win_start, win_len and the position of the one k-mer."""
    rng = np.random.default_rng(97 * k + b)
    texts, quals, at = [], [], []
    for length, j0 in ((150, 0), (150, 150 - k), (150, 71), (300, 123), (k + 1, 1), (k + 1, 0), (40, 17)):
        q = np.full(length, 33, np.uint8)                      # phred 0 everywhere ...
        q[j0:j0 + k] = 33 + b                                  # ... but k bases at exactly b: that k-mer sums to b*k,
        texts.append(LETTERS[rng.integers(0, 4, length)])      # its neighbours to b*(k-1) < b*k
        quals.append(q)
        at.append(j0)
    for phred in (40, 0):                                      # exactly k bases: one k-mer, good or not (:330-332)
        texts.append(LETTERS[rng.integers(0, 4, k)])
        quals.append(np.full(k, 33 + phred, np.uint8))
        at.append(0)
    lens = np.array([len(t) for t in texts], np.uint32)
    ws = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    return np.concatenate(texts), np.concatenate(quals), ws, lens, at


@pytest.mark.parametrize("p", [10, 20])
def test_oracle_sampling_with_exactly_one_good_kmer(p):
    """DOCUMENTED DEVIATION: with one good k-mer the reference's sampler keeps whatever the PREVIOUS
    window left in `samples` (stale positions indexing a 1-element vector: undefined behaviour; nothing at all for the
    first window of a run).  Oracle and product define the case instead: all p samples are that one k-mer."""
    k, b = 12, 25
    bases, quals, ws, wl, at = _single_good_kmer_batch(k, b)
    h, pos, has = oc.sample_windows(k, p, b * k, bases, quals, ws, wl)
    assert has.tolist() == [1] * len(ws)
    for w, j in enumerate(at):
        assert pos[w].tolist() == [j] * p, w
        want = int(onp.kmer_hashes(bases[int(ws[w]): int(ws[w]) + int(wl[w])], k)[j])
        assert h[w].tolist() == [want] * p, w
    # one more good k-mer and the sampler really samples: first half of the samples at the first, the rest at the second
    quals2 = quals.copy()
    quals2[int(ws[2]) + 100: int(ws[2]) + 100 + k] = 33 + b
    _, pos2, _ = oc.sample_windows(k, p, b * k, bases, quals2, ws, wl)
    assert sorted(set(pos2[2].tolist())) == [71, 100] and pos2[2, 0] == 71 and pos2[2, -1] == 100


@pytest.mark.gpu
@pytest.mark.parametrize("p", [10, 20])
def test_gpu_sampling_with_exactly_one_good_kmer(p):
    from bucket_map_amd import locate
    k, b = 12, 25
    bases, quals, ws, wl, at = _single_good_kmer_batch(k, b)
    want = oc.sample_windows(k, p, b * k, bases, quals, ws, wl)
    scan = locate.LocatorScan(k, p, 4, 6, 70000)
    got = scan.sample_windows(bases, quals, ws, wl, b * k)
    scan.close()
    for a, g, what in zip(want, got, ("hash", "position", "has-samples")):
        assert np.array_equal(a, g), what
    assert [got[1][w].tolist() for w in range(len(ws))] == [[j] * p for j in at]


@pytest.mark.gpu
@pytest.mark.parametrize("k,p,max_len", [(12, 10, 300), (12, 20, 300), (16, 10, 2500)])
def test_gpu_sampling_from_a_fastq_text_equals_oracle(k, p, max_len):
    """bml_sample_text_windows (what `bucketmap` calls with the memory-mapped FASTQ file): bases and qualities of a window
    lie apart in one buffer and the library gathers them piece by piece -- same samples as the oracle on the windows."""
    from bucket_map_amd import locate
    rng = np.random.default_rng(k * 100 + p)
    bases, quals, ws, wl = _sample_case(rng, 5000 if max_len <= 300 else 300, max_len, k)
    want = oc.sample_windows(k, p, 25 * k, bases, quals, ws, wl)
    # a FASTQ-like text: header, bases, "+", qualities per window
    parts, seq_at, qual_at, at = [], [], [], 0
    for w in range(len(ws)):
        b, q = bases[int(ws[w]): int(ws[w]) + int(wl[w])], quals[int(ws[w]): int(ws[w]) + int(wl[w])]
        head = np.frombuffer(f"@w{w}\n".encode(), np.uint8)
        parts += [head, b, np.frombuffer(b"\n+\n", np.uint8), q, np.frombuffer(b"\n", np.uint8)]
        seq_at.append(at + len(head))
        qual_at.append(at + len(head) + len(b) + 3)
        at += len(head) + 2 * len(b) + 4
    text = np.concatenate(parts)
    scan = locate.LocatorScan(k, p, 4, 6, 70000)
    got = scan.sample_text_windows(text, seq_at, qual_at, wl, 25 * k)
    for a, b, what in zip(want, got, ("hash", "position", "has-samples")):
        assert np.array_equal(a, b), what
    with pytest.raises(locate.BmlError):                       # a window past the end of the text
        scan.sample_text_windows(text, seq_at, np.full(len(ws), len(text), np.uint64), wl, 0)
    scan.close()


@pytest.mark.gpu
def test_gpu_genome_loaded_from_records_equals_the_flat_genome():
    """bml_load_genome_records (what the tool calls: no flattened copy of the genome on the host): records of very different
    sizes -- empty ones, one larger than two 32 MiB staging pieces -- concatenated on the device; the scan must find what it
    finds in the same genome uploaded as one string."""
    from bucket_map_amd import locate
    rng = np.random.default_rng(99)
    lens = [5, 0, 70_000, 1, (72 << 20) + 12_345, 0, 33_333, 900_000]
    records = [LETTERS[rng.integers(0, 4, n)] for n in lens]
    flat = np.concatenate(records)
    k, p, read_len, bucket_len = 12, 10, 150, 65536
    n_b = len(flat) // bucket_len
    bstart = (np.arange(n_b) * bucket_len).astype(np.uint64)
    blen = np.minimum(bucket_len + read_len, len(flat) - bstart.astype(np.int64)).astype(np.uint32)
    # reads cut from the genome, a few per region incl. the staging-piece boundaries at 32 and 64 MiB
    sh, sp, sl, pb, pw, pr = [], [], [], [], [], []
    for at in [100, 70_100, (32 << 20) - 70, (64 << 20) - 10, n_b * bucket_len - 400, (40 << 20) + 7]:
        seq = flat[at: at + read_len]
        hs = onp.kmer_hashes(seq, k)
        pos = [int(i) for i in onp.sample_positions(p, len(hs) - 1)]
        sh.append([int(hs[j]) for j in pos]); sp.append(pos); sl.append(read_len)
        pb.append(at // bucket_len); pw.append(len(sl) - 1); pr.append(0)
    order = np.argsort(pb, kind="stable")
    args = (np.array(sh, np.uint32), np.array(sp, np.uint16), np.array(sl, np.uint32), np.array(pb, np.uint32)[order],
            np.array(pw, np.uint32)[order], np.array(pr, np.uint8)[order])
    a = locate.LocatorScan(k, p, 4, 6, bucket_len + read_len)
    a.load_genome(flat, bstart, blen)
    want = a.locate(*args)
    a.close()
    b = locate.LocatorScan(k, p, 4, 6, bucket_len + read_len)
    b.load_genome_records(records, bstart, blen)
    got = b.locate(*args)
    b.close()
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
    assert (want[0] > 0).all() and (want[1] == p).all()        # every read found where it was cut, with all its samples


def _golden_sampling():
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    return json.load(open(os.path.join(here, "golden", "sampling_small.json")))["groups"]


def _sampling_batch(group):
    bases = np.frombuffer("".join(w["bases"] for w in group["windows"]).encode() + b"A", np.uint8)
    quals = np.frombuffer("".join(w["quals"] for w in group["windows"]).encode() + b"I", np.uint8)
    lens = [len(w["bases"]) for w in group["windows"]]
    ws = np.cumsum([0] + lens)[:-1].astype(np.uint64)
    want = (np.array([w["hash"] for w in group["windows"]], np.uint32), np.array([w["pos"] for w in group["windows"]], np.uint16),
            np.array([w["has"] for w in group["windows"]], np.uint8))
    return bases, quals, ws, np.array(lens, np.uint32), want


def test_oracle_sampling_matches_golden():
    """tests/golden/sampling_small.json comes from a plain-Python restatement (tests/golden/make_golden.py)."""
    for g in _golden_sampling():
        bases, quals, ws, wl, want = _sampling_batch(g)
        got = oc.sample_windows(g["k"], g["p"], g["min_base_quality"], bases, quals, ws, wl)
        for a, b, what in zip(want, got, ("hash", "position", "has-samples")):
            assert np.array_equal(a, b), (g["k"], g["p"], what)


@pytest.mark.gpu
def test_gpu_sampling_matches_golden():
    from bucket_map_amd import locate
    for g in _golden_sampling():
        bases, quals, ws, wl, want = _sampling_batch(g)
        scan = locate.LocatorScan(g["k"], g["p"], 4, 6, 70000)
        got = scan.sample_windows(bases, quals, ws, wl, g["min_base_quality"])
        for a, b, what in zip(want, got, ("hash", "position", "has-samples")):
            assert np.array_equal(a, b), (g["k"], g["p"], what)
        scan.close()


@pytest.mark.gpu
def test_gpu_sampling_errors():
    from bucket_map_amd import locate
    scan = locate.LocatorScan(12, 10, 4, 6, 70000)
    bases = np.frombuffer(b"ACGT" * 10, np.uint8)
    quals = np.full(40, ord("I"), np.uint8)
    with pytest.raises(locate.BmlError):                       # window past the end of the buffer
        scan.sample_windows(bases, quals, [30], [20], 0)
    h, pos, has = scan.sample_windows(bases, quals, np.zeros(0, np.uint64), np.zeros(0, np.uint32), 0)   # empty batch
    assert h.shape == (0, 10) and len(has) == 0
    h, pos, has = scan.sample_windows(bases, quals, [0, 5], [40, 0], 0)                                   # empty window
    assert has.tolist() == [1, 0] and not h[1].any()
    scan.close()
