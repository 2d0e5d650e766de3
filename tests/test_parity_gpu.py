"""Parity of the HIP path (through the C ABI, libbmf.so) with the CPU oracle.  Needs an MI355X.

Bar: bit-exact -- identical candidate counts and identical bucket ids in identical order, for both
orientations of every window.
"""
import json
import os

import numpy as np
import pytest

from conftest import Case, assert_same_candidates, oracle_map_windows

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _windows(case, n=None):
    import bucket_map_amd as bma
    rd = case.reads
    n = rd.n if n is None else min(n, rd.n)
    ws, wl, rid, _ = bma.windows_for_reads(rd.offsets[: n + 1], case.read_len)
    return ws, wl, rid


def _compare(case, n=None, what=""):
    import bucket_map_amd as bma
    rd = case.reads
    ws, wl, rid = _windows(case, n)
    ix = case.oracle_index()
    flt = case.gpu_filter()
    c_ref, b_ref, rows_ref = oracle_map_windows(ix, rd.bases, rd.quals, ws, wl)
    c_got, b_got = flt.map_windows(rd.bases, rd.quals, ws, wl)
    assert_same_candidates(c_ref, b_ref, c_got, b_got, what)
    # the packed form of the same call (what the host mapper uses): same counts, the lists back to back
    c_pk, ids = flt.map_windows_compact(rd.bases, rd.quals, ws, wl)
    assert np.array_equal(c_pk, c_ref)
    assert np.array_equal(ids, b_ref[np.arange(b_ref.shape[-1])[None, None, :] < c_ref[:, :, None]])
    # the opt-in early exit must not change a single output
    fe = case.gpu_filter(flags=bma.BMF_FLAG_EARLY_EXIT)
    c_e, b_e = fe.map_windows(rd.bases, rd.quals, ws, wl)
    assert_same_candidates(c_ref, b_ref, c_e, b_e, what + " (early exit)")
    fe.close()
    # ... nor may the two-pass form of it, forced here (the library picks it by index density)
    if case.cli["query_seed"] > case.cli["index_seed"] and case.num_buckets <= 65536:
        os.environ["BMF_PASS1_ROWS"] = "1"
        try:
            f2 = case.gpu_filter(flags=bma.BMF_FLAG_EARLY_EXIT)
        finally:
            del os.environ["BMF_PASS1_ROWS"]
        assert f2.info()["pass1_rows"] == 1
        c_e, b_e = f2.map_windows(rd.bases, rd.quals, ws, wl)
        assert_same_candidates(c_ref, b_ref, c_e, b_e, what + " (two-pass)")
        f2.close()
    # the device's row count is the unit of the algorithmic-bytes figure: must equal the oracle's
    batch = flt.batch(rd.bases, rd.quals, ws, wl)
    batch.run()
    assert batch.rows_anded() == rows_ref
    c2, b2 = batch.download()
    assert_same_candidates(c_ref, b_ref, c2, b2, what + " (batch)")
    batch.close()
    flt.close()
    # self-consistency: most reads find their source bucket (on the strand they were emitted on)
    hit = np.zeros(rd.n, bool)
    if len(ws):
        s = rd.truth_rc[rid].astype(np.int64)
        w = np.arange(len(ws))
        own = b_got[w, s]
        found = ((own == rd.truth_bucket[rid][:, None]) & (np.arange(own.shape[1])[None, :] < c_got[w, s][:, None])).any(axis=1)
        np.logical_or.at(hit, rid, found)
    n_reads = int(rid.max()) + 1 if len(rid) else 0
    return hit[:n_reads].mean() if n_reads else 1.0, c_got


def test_ecoli_like(ecoli_like):
    frac, _ = _compare(ecoli_like, what="ecoli-like")
    assert frac > 0.9


@pytest.mark.parametrize("nb_target,samples,err,k,q", [
    (700, 15, 0.4, 10, 7),        # 1 chunk per lane, partially filled
    (8192, 15, 0.4, 10, 7),       # exactly 64 chunks: CPL 1 full
    (8200, 15, 0.4, 10, 7),       # CPL 2, second round nearly empty
    (20000, 15, 0.4, 10, 7),      # CPL 3
    (26507, 15, 0.4, 10, 7),      # Egu geometry (CPL 4), P-default
    (26507, 20, 0.6, 12, 7),      # Egu geometry, P-bench (S=20, F=12, G=6)
    (46789, 15, 0.4, 10, 7),      # GRCh38 geometry (CPL 6)
    (65536, 15, 0.4, 9, 7),       # largest single-wave NB (CPL 8)
    (65537, 15, 0.4, 9, 7),       # 2 slices, the second holds one bucket
    (140000, 15, 0.4, 10, 7),     # 3 slices (NB > 65 536: one wave per 65 536-bucket slice + merge)
    (140000, 20, 0.6, 12, 7),     # 3 slices, P-bench
])
def test_geometries(nb_target, samples, err, k, q):
    bucket_len = 256
    case = Case(record_lengths=[nb_target * bucket_len - 17], bucket_len=bucket_len, read_len=100, n_reads=20_000, q=q,
                k=k, samples=samples, error_rate=err, sub=0.01, seed=20240100 + nb_target)
    assert case.num_buckets == nb_target
    frac, _ = _compare(case, what=f"NB={nb_target} S={samples}")
    assert frac > 0.85


def test_fracminhash_index_has_unindexed_qgrams():
    # -f 0.25: most q-grams are not indexed (kmer_to_index == -1), reverse-complement samples can have
    # no indexed q-gram at all -> all-ones vector
    case = Case(record_lengths=[300_000, 41_000], bucket_len=1024, read_len=120, n_reads=300, q=7, k=10,
                kmer_frac=0.25, extra_buckets=3, seed=77)
    assert (case.index.kmer_to_index() < 0).mean() > 0.5
    _compare(case, what="kmer_frac=0.25")


def test_noisy_qualities_and_quality_filter():
    case = Case(record_lengths=[200_000], bucket_len=2048, read_len=150, n_reads=300, noisy_quals=True,
                base_quality=37, seed=78)
    _, counts = _compare(case, what="noisy quals")
    rejected = (counts.sum(axis=1) == 0).mean()
    assert 0.02 < rejected < 0.98               # the quality filter rejects some reads, not all


def test_long_reads_are_cut_into_five_windows():
    # reads longer than 2*read_len -> 5 overlapping windows at Sampler(5) positions (q_gram_mapper.h:512-516)
    case = Case(record_lengths=[400_000], bucket_len=8192, read_len=150, n_reads=60, sim_read_len=1000, sub=0.01,
                seed=79)
    ws, wl, rid = _windows(case)
    assert len(ws) == 5 * case.reads.n and (wl <= 150).all()
    frac, _ = _compare(case, what="long reads")
    assert frac > 0.9


@pytest.mark.parametrize("read_len,expect_bitmap_lds", [(2000, True), (9000, True), (16384, False)])
def test_sample_kernel_lds_regimes(read_len, expect_bitmap_lds):
    """The sample kernel sizes its workgroups by -r: 16 waves beside the 32 KiB q-gram bitmap for short reads, fewer
    for long ones, and at -r 16384 one wave per workgroup with the bitmap left in L2.  Same outputs in every regime,
    with windows that start at every byte alignment and end anywhere."""
    import bucket_map_amd as bma
    from oracle import oracle_c
    case = Case(record_lengths=[3_000_000], bucket_len=65536, read_len=read_len, n_reads=64, sim_read_len=read_len, sub=0.01,
                ins=0.002, dele=0.002, seed=90 + read_len)
    rd = case.reads
    rng = np.random.default_rng(read_len)
    # whole reads, plus windows cut at random offsets and lengths inside them (every alignment of the 8-byte loads)
    ws = [int(rd.offsets[r]) for r in range(rd.n)]
    wl = [min(read_len, int(rd.offsets[r + 1] - rd.offsets[r])) for r in range(rd.n)]
    for r in range(rd.n):
        n = int(rd.offsets[r + 1] - rd.offsets[r])
        a = int(rng.integers(0, max(1, n - 20)))
        ws.append(int(rd.offsets[r]) + a)
        wl.append(int(rng.integers(0, min(read_len, n - a) + 1)))
    ws, wl = np.array(ws, np.uint64), np.array(wl, np.uint32)
    ix = case.oracle_index()
    c_ref, b_ref, rows_ref = oracle_map_windows(ix, rd.bases, rd.quals, ws, wl)
    for flags in (0, bma.BMF_FLAG_EARLY_EXIT):
        flt = case.gpu_filter(flags=flags)
        c, b = flt.map_windows(rd.bases, rd.quals, ws, wl)
        assert_same_candidates(c_ref, b_ref, c, b, f"-r {read_len} flags {flags}")
        flt.close()
    assert (c_ref.sum(axis=1) > 0).mean() > 0.5
    # the regime this -r was meant to exercise (bmf_create's sizing: packed bases, prefix sums, good k-mers per wave)
    up16 = lambda x: (x + 15) & ~15
    stream = up16(read_len + 14)
    per_wave = up16(up16((stream // 16 + 3) * 4) + up16((stream + 1) * 4) + 4 * (read_len - 12 + 1))
    assert (32768 + per_wave <= 159 * 1024) == expect_bitmap_lds


def test_folded_first_pass_and_its_fallback_guard():
    """The pruning kernels' first pass normally streams a folded copy of the index (one bit per 2 or 4 buckets), chosen
    by a model of how many chunks survive it by chance; a measured guard switches a context back to the unfolded
    passes when the model turns out wrong.  Both forms, and the switch itself, give the oracle's outputs."""
    import bucket_map_amd as bma
    case = Case(record_lengths=[26507 * 256 - 17], bucket_len=256, read_len=100, n_reads=6000, q=7, k=10, sub=0.01, seed=20240200)
    rd = case.reads
    ws, wl, _ = _windows(case)
    c_ref, b_ref, _ = oracle_map_windows(case.oracle_index(), rd.bases, rd.quals, ws, wl)
    flt = case.gpu_filter(flags=bma.BMF_FLAG_EARLY_EXIT)
    assert flt.info()["pass1_fold"] in (2, 4) and flt.info()["pass1_fold_rows"] >= 1
    c, b = flt.map_windows(rd.bases, rd.quals, ws, wl)
    assert_same_candidates(c_ref, b_ref, c, b, "folded first pass")
    os.environ["BMF_GUARD_TRIP"] = "1"                  # pretend the last run overflowed the recount kernel's lanes
    try:
        c, b = flt.map_windows(rd.bases, rd.quals, ws, wl)   # this call notices and falls back ...
    finally:
        del os.environ["BMF_GUARD_TRIP"]
    assert_same_candidates(c_ref, b_ref, c, b, "the call that falls back")
    assert flt.info()["pass1_fold"] == 1                    # ... for good
    c, b = flt.map_windows(rd.bases, rd.quals, ws, wl)
    assert_same_candidates(c_ref, b_ref, c, b, "after the fallback")
    # the fallback restores the WHOLE unfolded choice -- rows of the first pass and the recount kernel's lanes per item --
    # so the context now does exactly what one that never had a folded copy does: same items through the recount and the
    # slow kernels
    batch = flt.batch(rd.bases, rd.quals, ws, wl)
    batch.run()
    after_trip = batch.pass2_counts()
    batch.close()
    flt.close()
    os.environ["BMF_FOLD"] = "0"
    try:
        plain2 = case.gpu_filter(flags=bma.BMF_FLAG_EARLY_EXIT)
    finally:
        del os.environ["BMF_FOLD"]
    batch = plain2.batch(rd.bases, rd.quals, ws, wl)
    batch.run()
    assert batch.pass2_counts() == after_trip
    batch.close()
    plain2.close()


def test_pruning_form_is_measured_on_the_first_large_batch():
    """tune_pruned: the first batch of at least BMF_TUNE_WINDOWS windows times every pruning form on its own first
    windows and keeps the fastest; every form writes the oracle's outputs, before, while and after.  On a genome-like
    genome (rows of very different densities, reads in repeats) so that the slow path and the level rounds all run."""
    import bucket_map_amd as bma
    case = Case(record_lengths=[9_000_000], bucket_len=1024, read_len=100, n_reads=9000, q=7, k=10, sub=0.01, seed=20240300,
                profile="genome")
    rd = case.reads
    ws, wl, _ = _windows(case)
    c_ref, b_ref, _ = oracle_map_windows(case.oracle_index(), rd.bases, rd.quals, ws, wl)
    assert 0.001 < (c_ref.sum(axis=1) == 0).mean() < 0.9 and (c_ref.max(axis=1) > 1).mean() > 0.01   # repeats: ties, rejections
    os.environ["BMF_TUNE_WINDOWS"] = "4096"
    try:
        flt = case.gpu_filter(flags=bma.BMF_FLAG_EARLY_EXIT)
    finally:
        del os.environ["BMF_TUNE_WINDOWS"]
    before = flt.info()
    c, b = flt.map_windows(rd.bases[: int(rd.offsets[1000])], rd.quals[: int(rd.offsets[1000])], ws[:1000], wl[:1000])
    assert_same_candidates(c_ref[:1000], b_ref[:1000], c, b, "small batch: the model's choice")
    assert flt.info() == before
    c, b = flt.map_windows(rd.bases, rd.quals, ws, wl)                       # large enough: this call measures
    assert_same_candidates(c_ref, b_ref, c, b, "the batch that tunes")
    chosen = flt.info()
    c, b = flt.map_windows(rd.bases, rd.quals, ws, wl)
    assert_same_candidates(c_ref, b_ref, c, b, "after tuning")
    assert flt.info() == chosen                                              # measured once
    flt.close()


def test_repeats_overflow_max_candidates():
    # 40 identical records -> a read matches > 30 buckets equally well -> list cleared (q_gram_mapper.h:471-476)
    from bucket_map_amd import host
    g1 = host.Genome.synth(5, [3000])
    seq = bytes(g1.record_seq(0))
    path = "/tmp/bm_repeat_genome.fa"
    with open(path, "w") as f:
        for i in range(40):
            f.write(f">rep{i}\n{seq.decode()}\n")
    case = Case.__new__(Case)
    case.genome = host.Genome.read_fasta(path)
    case.bucket_len, case.read_len = 4096, 100
    case.num_buckets = case.genome.awk_bucket_num(4096)
    case.index = host.Index(case.genome, case.num_buckets, 4096, 100, q=7)
    case.reads = host.Reads(case.genome, 4096, 100, 100, 100, seed=9)
    case.cli = dict(index_seed=7, query_seed=10, read_len=100, mapper_samples=15, max_error_rate=0.4,
                    distinguishability=0.0, average_base_quality=25)
    _, counts = _compare(case, what="repeats")
    assert counts.max() == 0


@pytest.mark.parametrize("copies,expect_cleared", [(3, False), (40, True)])
def test_ties_across_slices(copies, expect_cleared):
    # NB > 65 536: identical records far apart put equally good buckets into DIFFERENT 65 536-bucket slices;
    # the merge must concatenate them in ascending order, and clear the list when the slices together
    # exceed max_candidates although no single slice does (q_gram_mapper.h:471-476).
    from bucket_map_amd import host
    per_copy = 140_000 // copies
    g1 = host.Genome.synth(6, [per_copy * 256])          # keep alive: record_seq is a view into it
    unit = bytes(g1.record_seq(0)).decode()
    path = f"/tmp/bm_slices_{copies}.fa"
    with open(path, "w") as f:
        for i in range(copies):
            f.write(f">copy{i}\n{unit}\n")
    case = Case.__new__(Case)
    case.genome = host.Genome.read_fasta(path)
    case.bucket_len, case.read_len = 256, 100
    case.num_buckets = case.genome.awk_bucket_num(256)
    assert case.num_buckets > 2 * 65536
    case.index = host.Index(case.genome, case.num_buckets, 256, 100, q=7)
    case.reads = host.Reads(case.genome, 256, 100, 100, 200, sub=0.0, ins=0.0, dele=0.0, seed=10)
    case.cli = dict(index_seed=7, query_seed=10, read_len=100, mapper_samples=15, max_error_rate=0.4,
                    distinguishability=0.0, average_base_quality=25)
    _, counts = _compare(case, what=f"{copies} copies")
    if expect_cleared:
        assert counts.max() == 0
    else:
        assert (counts.max(axis=1) >= copies).mean() > 0.8    # one candidate per copy (plus overlap neighbours)


def test_ragged_and_degenerate_windows(ecoli_like):
    case = ecoli_like
    rd = case.reads
    rng = np.random.default_rng(5)
    pieces_b, pieces_q, off = [], [], [0]
    for r in range(120):
        o0, o1 = int(rd.offsets[r]), int(rd.offsets[r + 1])
        b, q = rd.bases[o0:o1][:150].copy(), rd.quals[o0:o1][:150].copy()
        mode = r % 8
        if mode == 1:
            b, q = b[:0], q[:0]                                 # empty window
        elif mode == 2:
            b, q = b[:11], q[:11]                               # shorter than k
        elif mode == 3:
            b, q = b[:12], q[:12]                               # exactly one k-mer
        elif mode == 4:
            n = int(rng.integers(13, len(b)))
            b, q = b[:n], q[:n]                                 # ragged
        elif mode == 5:
            q[:] = ord("#")                                     # all low quality
        elif mode == 6:
            b[20:60] = ord("N")                                 # ambiguity codes
        elif mode == 7:
            b = np.frombuffer(bytes(b).lower(), np.uint8).copy()  # lower case
        pieces_b.append(b); pieces_q.append(q); off.append(off[-1] + len(b))
    bases, quals, off = np.concatenate(pieces_b), np.concatenate(pieces_q), np.array(off, np.uint64)
    ws, wl = off[:-1], np.diff(off).astype(np.uint32)
    ix, flt = case.oracle_index(), case.gpu_filter()
    c_ref, b_ref, _ = ix.map_windows(bases, quals, ws, wl)
    c_got, b_got = flt.map_windows(bases, quals, ws, wl)
    assert_same_candidates(c_ref, b_ref, c_got, b_got, "ragged")
    # zero windows is a no-op
    c0, _ = flt.map_windows(bases, quals, ws[:0], wl[:0])
    assert c0.shape == (0, 2)
    # overlapping views of one buffer
    ws2 = np.array([0, 10, 20, 5], np.uint64)
    wl2 = np.array([150, 140, 100, 150], np.uint32)
    c_ref, b_ref, _ = ix.map_windows(rd.bases, rd.quals, ws2, wl2)
    c_got, b_got = flt.map_windows(rd.bases, rd.quals, ws2, wl2)
    assert_same_candidates(c_ref, b_ref, c_got, b_got, "overlapping views")
    flt.close()


@pytest.mark.parametrize("flags", [0, 1])
def test_windows_gathered_from_a_fastq_text(ecoli_like, flags, monkeypatch):
    """bmf_map_text_windows_compact (what `bucketmap` calls with the memory-mapped FASTQ file): bases and qualities of a
    window lie apart in one buffer and the library gathers them piece by piece, by several threads -- same counts and ids
    as bmf_map_windows_compact on the same windows laid out back to back; windows out of order, overlapping, empty,
    and pieces smaller than the thread count's share included."""
    case = ecoli_like
    rd = case.reads
    rng = np.random.default_rng(77)
    text, seq_at, qual_at, lens = [], [], [], []
    at = 0
    for r in range(rd.n):
        o0, o1 = int(rd.offsets[r]), int(rd.offsets[r + 1])
        b, q = rd.bases[o0:o1], rd.quals[o0:o1]
        head = np.frombuffer(f"@read{r} some description\n".encode(), np.uint8)
        text += [head, b, np.frombuffer(b"\n+\n", np.uint8), q, np.frombuffer(b"\n", np.uint8)]
        seq_at.append(at + len(head))
        qual_at.append(at + len(head) + len(b) + 3)
        lens.append(min(len(b), case.read_len))
        at += len(head) + 2 * len(b) + 4
    text = np.concatenate(text)
    seq_at, qual_at, lens = np.array(seq_at, np.uint64), np.array(qual_at, np.uint64), np.array(lens, np.uint32)
    # a few windows again, shifted and shortened (overlapping views), an empty one, and everything shuffled
    extra = rng.integers(0, rd.n, 40)
    seq_at = np.concatenate([seq_at, seq_at[extra] + 7, seq_at[:1]])
    qual_at = np.concatenate([qual_at, qual_at[extra] + 7, qual_at[:1]])
    lens = np.concatenate([lens, lens[extra] - 7, np.zeros(1, np.uint32)]).astype(np.uint32)
    perm = rng.permutation(len(lens))
    seq_at, qual_at, lens = seq_at[perm], qual_at[perm], lens[perm]
    # reference layout: the same windows back to back
    ws = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    bases = np.concatenate([text[int(a): int(a) + int(n)] for a, n in zip(seq_at, lens)] + [np.zeros(1, np.uint8)])
    quals = np.concatenate([text[int(a): int(a) + int(n)] for a, n in zip(qual_at, lens)] + [np.zeros(1, np.uint8)])
    f = case.gpu_filter(flags=flags)
    want_c, want_ids = f.map_windows_compact(bases, quals, ws, lens)
    want_ids = want_ids.copy()
    for threads, piece in (("1", "100000"), ("5", "64"), ("3", "97")):
        monkeypatch.setenv("BMF_PIECE_WINDOWS", piece)
        monkeypatch.setenv("BMF_GATHER_THREADS", threads)
        got_c, got_ids = f.map_text_windows_compact(text, seq_at, qual_at, lens)
        assert np.array_equal(got_c, want_c) and np.array_equal(got_ids, want_ids), (threads, piece)
    assert want_c.sum() > 0.9 * rd.n
    # a window past the end of the text is an error, not a read
    import bucket_map_amd as bma
    with pytest.raises(bma.BmfError):
        f.map_text_windows_compact(text, seq_at, np.full_like(qual_at, len(text)), lens)
    f.close()


def test_golden_reads_on_gpu():
    import bucket_map_amd as bma
    with open(os.path.join(GOLDEN, "reads_small.json")) as f:
        g = json.load(f)
    p = bma.Params(num_buckets=g["num_buckets"], q=g["q"], k=g["k"], num_samples=g["S"], num_fault=g["F"],
                   threshold=g["threshold"], min_base_quality=g["min_base_quality"], read_len=g["read_len"])
    flt = bma.Filter(p)
    flt.load_index(np.array(g["rows"], np.uint8), np.array(g["kmer_to_index"], np.int32))
    bases = np.frombuffer("".join(r["bases"] for r in g["reads"]).encode(), np.uint8)
    quals = np.frombuffer("".join(r["quals"] for r in g["reads"]).encode(), np.uint8)
    off = np.cumsum([0] + [len(r["bases"]) for r in g["reads"]]).astype(np.uint64)
    counts, buckets = flt.map_windows(bases, quals, off[:-1], np.diff(off).astype(np.uint32))
    for i, r in enumerate(g["reads"]):
        assert list(buckets[i, 0, : counts[i, 0]]) == r["fwd"], i
        assert list(buckets[i, 1, : counts[i, 1]]) == r["rc"], i
    flt.close()


def test_golden_tiny_index_on_gpu():
    # Drive the vote with chosen sample hashes: k=3 windows of exactly 3 bases hold one k-mer = one
    # sample (S=1), so every k-mer of the exhaustive S=1 fixture becomes one window.
    import bucket_map_amd as bma
    from oracle import oracle_c as oc
    with open(os.path.join(GOLDEN, "tiny_index.json")) as f:
        g = json.load(f)
    case = g["cases"][0]
    assert case["S"] == 1
    rows, k2i = np.array(g["rows"], np.uint8), np.array(g["kmer_to_index"], np.int32)
    kw = dict(q=g["q"], k=g["k"], num_samples=1, num_fault=case["F"], threshold=0, min_base_quality=0, read_len=8,
              max_candidates=64)
    flt = bma.Filter(bma.Params(num_buckets=g["num_buckets"], **kw))
    flt.load_index(rows, k2i)
    letters = b"ACGT"
    wins = [bytes(letters[(h >> (2 * (g["k"] - 1 - i))) & 3] for i in range(g["k"])) for (h,) in case["hashes"]]
    bases = np.frombuffer(b"".join(wins), np.uint8)
    quals = np.full(len(bases), ord("E"), np.uint8)
    ws = (np.arange(len(wins)) * g["k"]).astype(np.uint64)
    wl = np.full(len(wins), g["k"], np.uint32)
    counts, buckets = flt.map_windows(bases, quals, ws, wl)
    # a k-mer none of whose q-grams is indexed is not "good": the window is rejected, and the oracle
    # says so too -- so compare through the oracle first
    ix = oc.Index(oc.make_params(g["num_buckets"], **kw), rows, k2i)
    c_ref, b_ref, _ = ix.map_windows(bases, quals, ws, wl)
    assert_same_candidates(c_ref, b_ref, counts, buckets, "tiny index")
    # and where the window was accepted, the forward list is the committed expectation
    checked = 0
    for i, want in enumerate(case["expected"]):
        if c_ref[i].sum() and len(want) <= 64:
            assert list(buckets[i, 0, : counts[i, 0]]) == want
            checked += 1
    assert checked > 30
    flt.close()


def test_error_behaviour():
    import bucket_map_amd as bma
    p = bma.Params.from_cli(100, read_len=50)
    flt = bma.Filter(p)
    bases = np.frombuffer(b"ACGT" * 20, np.uint8)
    quals = np.full(80, ord("E"), np.uint8)
    ws, wl = np.array([0], np.uint64), np.array([40], np.uint32)
    with pytest.raises(bma.BmfError) as e:           # query before load: q_gram_mapper.h:389-393
        flt.map_windows(bases, quals, ws, wl)
    assert e.value.code == bma.BMF_ERR_STATE
    rows = np.zeros((4 ** 9, 13), np.uint8)
    flt.load_index(rows, np.arange(4 ** 9, dtype=np.int32))
    with pytest.raises(bma.BmfError) as e:           # second load: q_gram_mapper.h:325-328
        flt.load_index(rows, np.arange(4 ** 9, dtype=np.int32))
    assert e.value.code == bma.BMF_ERR_STATE
    with pytest.raises(bma.BmfError) as e:           # window longer than read_len
        flt.map_windows(bases, quals, ws, np.array([80], np.uint32))
    assert e.value.code == bma.BMF_ERR_ARG
    with pytest.raises(bma.BmfError) as e:           # window outside the buffer
        flt.map_windows(bases, quals, np.array([60], np.uint64), wl)
    assert e.value.code == bma.BMF_ERR_ARG
    flt.reset()                                      # mapper::reset frees the index, context stays usable
    flt.load_index(rows, np.arange(4 ** 9, dtype=np.int32))
    c, _ = flt.map_windows(bases, quals, ws, wl)
    assert c.sum() == 0                              # all-zero index: every bucket misses every sample
    flt.close()
    with pytest.raises(bma.BmfError):
        bma.Filter(bma.Params.from_cli(17_000_000))  # NB > 16 777 216 unsupported
    # packed output with too little room for the ids: an error, not an overrun
    flt = bma.Filter(p)
    rows[:] = 0xFF                                   # every bucket hits every sample: 2 x 30 ids per window
    flt.load_index(rows, np.arange(4 ** 9, dtype=np.int32))
    flt2 = bma.Filter(bma.Params(num_buckets=20, read_len=50))
    flt2.load_index(np.full((4 ** 9, 3), 0xFF, np.uint8), np.arange(4 ** 9, dtype=np.int32))
    c, ids = flt2.map_windows_compact(bases, quals, ws, wl)
    assert c.tolist() == [[20, 20]] and ids.tolist() == list(range(20)) * 2
    with pytest.raises(bma.BmfError) as e:
        flt2.map_windows_compact(bases, quals, ws, wl, out=(np.zeros((1, 2), np.uint32), np.zeros(39, np.uint32)))
    assert e.value.code == bma.BMF_ERR_ARG
    c, ids = flt.map_windows_compact(bases, quals, ws, wl)   # 100 buckets tie: more than 30 -> cleared, nothing to pack
    assert c.sum() == 0 and len(ids) == 0
    flt.close(); flt2.close()


def test_zeros_match_oracle(ecoli_like):
    ix, flt = ecoli_like.oracle_index(), ecoli_like.gpu_filter()
    assert np.array_equal(flt.zeros(), ix.zeros())
    flt.close()


def test_determinism_and_batch_invariance(ecoli_like):
    case = ecoli_like
    rd = case.reads
    ws, wl, _ = _windows(case)
    flt = case.gpu_filter()
    c1, b1 = flt.map_windows(rd.bases, rd.quals, ws, wl)
    c2, b2 = flt.map_windows(rd.bases, rd.quals, ws, wl)
    assert_same_candidates(c1, b1, c2, b2, "idempotence")
    # splitting the batch must not change any window's result
    h = len(ws) // 3
    ca, ba = flt.map_windows(rd.bases, rd.quals, ws[:h], wl[:h])
    cb, bb = flt.map_windows(rd.bases, rd.quals, ws[h:], wl[h:])
    assert_same_candidates(c1, b1, np.concatenate([ca, cb]), np.concatenate([ba, bb]), "batch split")
    # permuting the windows permutes the results
    perm = np.random.default_rng(1).permutation(len(ws))
    cp, bp = flt.map_windows(rd.bases, rd.quals, ws[perm], wl[perm])
    assert_same_candidates(c1[perm], b1[perm], cp, bp, "permutation")
    flt.close()


def test_pass2_counts_report_the_two_pass_kernels(ecoli_like):
    """bmf_batch_pass2_counts: zero unless the two-pass pruning kernels served the run; with them forced on,
    recounted + slow-path items never exceed the items, and the outputs equal the default kernel's."""
    import bucket_map_amd as bma
    case = ecoli_like
    rd = case.reads
    ws, wl, _ = _windows(case)
    plain = case.gpu_filter()
    b = plain.batch(rd.bases, rd.quals, ws, wl)
    b.run()
    assert b.pass2_counts() == (0, 0)
    want = b.download()
    b.close(); plain.close()
    os.environ["BMF_PASS1_ROWS"] = "1"
    try:
        f2 = case.gpu_filter(flags=bma.BMF_FLAG_EARLY_EXIT)
    finally:
        del os.environ["BMF_PASS1_ROWS"]
    b = f2.batch(rd.bases, rd.quals, ws, wl)
    b.run()
    recounted, slow = b.pass2_counts()
    assert 0 < recounted + slow <= 2 * len(ws)
    got = b.download()
    assert_same_candidates(want[0], want[1], got[0], got[1], "two-pass")
    b.close(); f2.close()
