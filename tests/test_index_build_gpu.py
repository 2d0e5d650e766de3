"""GPU index build (bmf_build_index) against the host indexer (bm::build_index, itself checked against a
numpy brute force in tests/test_host.py): the rows must be byte-identical, and a filter whose index was
built on the GPU must answer exactly like one that loaded the host-built rows."""
import numpy as np
import pytest

from conftest import Case, assert_same_candidates

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", [
    ("small, NB crosses a word", dict(record_lengths=[150_000, 9_000, 700], bucket_len=2048, read_len=150, q=9)),
    ("three records, padding buckets", dict(record_lengths=[300_000, 41_000, 41_500], bucket_len=1024, read_len=120, q=7, k=10, extra_buckets=3)),
    ("FracMinHash 0.25", dict(record_lengths=[200_000], bucket_len=1024, read_len=120, q=7, k=10, kmer_frac=0.25)),
    ("q=10", dict(record_lengths=[500_000], bucket_len=4096, read_len=150, q=10, k=12)),
    ("q=3", dict(record_lengths=[20_000], bucket_len=512, read_len=60, q=3, k=6)),
    ("NB 26507 geometry", dict(record_lengths=[26507 * 256 - 17], bucket_len=256, read_len=100, q=7, k=10)),
])
def test_rows_identical_to_host_indexer(name, kw):
    import bucket_map_amd as bma
    case = Case(n_reads=200, **kw)
    g = case.genome
    flat, _ = g.flat()
    bstart, blen = g.bucket_views(case.bucket_len, case.read_len)
    k2i = case.index.kmer_to_index()
    flt = bma.Filter(bma.Params.from_cli(case.num_buckets, **case.cli))
    flt.build_index(flat, bstart, blen, k2i)
    rows = flt.index_download()
    assert rows.shape == case.index.rows().shape
    assert np.array_equal(rows, case.index.rows()), name
    # and it answers queries exactly like a filter that loaded the host-built rows
    ref = case.gpu_filter()
    rd = case.reads
    ws, wl, _, _ = bma.windows_for_reads(rd.offsets, case.read_len)
    c1, b1 = ref.map_windows(rd.bases, rd.quals, ws, wl)
    c2, b2 = flt.map_windows(rd.bases, rd.quals, ws, wl)
    assert_same_candidates(c1, b1, c2, b2, name)
    assert np.array_equal(flt.zeros(), ref.zeros())
    flt.close(); ref.close()


def test_build_errors():
    import bucket_map_amd as bma
    from bucket_map_amd import host
    g = host.Genome.synth(1, [10_000])
    flat, _ = g.flat()
    bs, bl = g.bucket_views(1024, 100)
    flt = bma.Filter(bma.Params.from_cli(len(bs) - 1, read_len=100, index_seed=7, query_seed=10))
    with pytest.raises(bma.BmfError):                     # more buckets than NB
        flt.build_index(flat, bs, bl, host.select_qgrams(7))
    flt.close()
    flt = bma.Filter(bma.Params.from_cli(len(bs), read_len=100, index_seed=7, query_seed=10))
    with pytest.raises(bma.BmfError):                     # wrong table size
        flt.build_index(flat, bs, bl, host.select_qgrams(6))
    bad = bs.copy(); bad[-1] = len(flat)
    with pytest.raises(bma.BmfError):                     # bucket outside the genome
        flt.build_index(flat, bad, bl, host.select_qgrams(7))
    flt.build_index(flat, bs, bl, host.select_qgrams(7))
    with pytest.raises(bma.BmfError):                     # already loaded
        flt.build_index(flat, bs, bl, host.select_qgrams(7))
    flt.close()


def test_transpose_beyond_2_pow_32_lanes():
    """q = 10 with 300 000 tiny buckets: (groups of 64 buckets) x (4^q / 64 q-gram words) x 64 lanes = 4.9e9 > 2^32.
    A 32-bit wave index aliases there and leaves rows unwritten (all-zero rows = false-negative buckets).  The rows
    (39 GB) stay in HBM; their popcounts -- zeros[row] = NB - #buckets holding the q-gram -- are checked for ALL
    4^q rows against a numpy count over the same buckets."""
    import bucket_map_amd as bma
    from bucket_map_amd import host
    q, bucket_len, read_len, nb = 10, 16, 12, 300_000
    g = host.Genome.synth(77, [nb * bucket_len + 5])
    flat, _ = g.flat()
    bstart, blen = g.bucket_views(bucket_len, read_len)
    assert len(bstart) == nb and g.awk_bucket_num(bucket_len) == nb + 1
    k2i = host.select_qgrams(q)
    flt = bma.Filter(bma.Params.from_cli(nb + 1, read_len=read_len, index_seed=q, query_seed=q))
    flt.build_index(flat, bstart, blen, k2i)
    zeros = flt.zeros()
    flt.close()
    # numpy: hash of every q-gram start, then per bucket the distinct hashes it holds
    codes = np.zeros(256, np.uint8)
    for c, r in zip(b"ACGT", range(4)):
        codes[c] = r
    r = codes[flat].astype(np.uint32)
    n = len(r) - q + 1
    h = np.zeros(n, np.uint32)
    for t in range(q):
        h = h * 4 + r[t:t + n]
    per = int(blen.max()) - q + 1
    pos = bstart.astype(np.int64)[:, None] + np.arange(per)[None, :]
    ok = np.arange(per)[None, :] < (blen.astype(np.int64) - q + 1)[:, None]
    hb = np.where(ok, h[np.minimum(pos, n - 1)], np.uint32(0xFFFFFFFF))
    hb.sort(axis=1)
    first = np.ones_like(hb, bool)
    first[:, 1:] = hb[:, 1:] != hb[:, :-1]
    keep = first & (hb != 0xFFFFFFFF)
    counts = np.bincount(hb[keep].astype(np.int64), minlength=4 ** q)
    assert counts.max() > 0 and (counts > 0).mean() > 0.9
    assert np.array_equal(zeros, (nb + 1 - counts).astype(np.uint32))
