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
