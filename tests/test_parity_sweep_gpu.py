"""Parameter sweep of the filter's parity: random small indexes and reads under many (q, k, S, F,
max_candidates, kmer_frac, NB) combinations, GPU vs oracle, bit-exact.  Covers the corners of the kernel
variant table the fixed geometries do not: PLANES 2..5 (F = 1..31), G = 1..8, S = 1..64, windows with
fewer good k-mers than samples (repeated sample positions), thresholds that reject k-mers."""
import os

import numpy as np
import pytest

from conftest import assert_same_candidates

pytestmark = pytest.mark.gpu


def random_case(rng, *, nb, q, k, S, F, max_cand, kmer_frac, density, read_len, n_reads, threshold, minq, skew=False):
    import bucket_map_amd as bma
    from oracle import oracle_c as oc
    n_q = 4 ** q
    kept = rng.random(n_q) < kmer_frac
    k2i = np.full(n_q, -1, np.int32)
    k2i[kept] = np.arange(kept.sum())
    # rows of one density, or -- as on a real genome -- rows whose densities differ widely, over buckets some of which hold
    # nearly every q-gram (the pruning kernels then see every level between "the read's own bucket" and "unrelated")
    p_bit = np.full((int(kept.sum()), nb), density)
    if skew:
        p_bit = p_bit * (rng.random((p_bit.shape[0], 1)) ** 2 * 2.0)
        hot = rng.random(nb) < 0.02
        p_bit[:, hot] = np.maximum(p_bit[:, hot], rng.choice([0.6, 0.9, 0.99], (1, int(hot.sum()))))
    bits = rng.random(p_bit.shape) < np.minimum(p_bit, 1.0)
    rows = np.packbits(bits, axis=1, bitorder="little")
    letters = np.frombuffer(b"ACGTacgtN", np.uint8)
    lens = rng.integers(0, read_len + 1, n_reads)
    lens[: n_reads // 2] = read_len
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    bases = letters[rng.integers(0, len(letters), int(off[-1]))]
    quals = rng.integers(33, 33 + 42, int(off[-1])).astype(np.uint8)
    kw = dict(q=q, k=k, num_samples=S, num_fault=F, threshold=threshold, min_base_quality=minq, max_candidates=max_cand,
              read_len=read_len)
    ix = oc.Index(oc.make_params(nb, **kw), rows, k2i)
    flt = bma.Filter(bma.Params(num_buckets=nb, **kw))
    flt.load_index(rows, k2i)
    fe = bma.Filter(bma.Params(num_buckets=nb, flags=bma.BMF_FLAG_EARLY_EXIT, **kw))
    fe.load_index(rows, k2i)
    ws, wl = off[:-1], lens.astype(np.uint32)
    c_ref, b_ref, rows_ref = ix.map_windows(bases, quals, ws, wl)
    c_got, b_got = flt.map_windows(bases, quals, ws, wl)
    c_e, b_e = fe.map_windows(bases, quals, ws, wl)
    flt.close(); fe.close()
    out = [c_ref, b_ref, c_got, b_got, c_e, b_e]
    # the single-pass pruning kernel, forced (the library keeps the plain kernel for short rows and sparse indexes)
    os.environ["BMF_PASS1_ROWS"] = "0"
    try:
        f1 = bma.Filter(bma.Params(num_buckets=nb, flags=bma.BMF_FLAG_EARLY_EXIT, **kw))
        f1.load_index(rows, k2i)
    finally:
        del os.environ["BMF_PASS1_ROWS"]
    c_1, b_1 = f1.map_windows(bases, quals, ws, wl)
    f1.close()
    assert_same_candidates(c_ref, b_ref, c_1, b_1, "single-pass pruning forced")
    if k > q:
        # the two-pass pruning kernel, forced (the library only picks it for sparse indexes), r rows in pass 1
        # and either form of the recount kernel (16 or 32 lanes per item), whatever the density model would pick
        # and either form of its first pass: r rows of the index itself, or r rows of the index folded 2 or 4 buckets to a bit
        fold = int(rng.choice([0, 2, 4]))
        os.environ["BMF_PASS1_ROWS"] = str(int(rng.integers(1, k - q + 1)))
        os.environ["BMF_MAX_LIVE"] = str(int(rng.choice([16, 32])))
        os.environ["BMF_FOLD"] = str(fold)
        os.environ["BMF_FOLD_ROWS"] = str(int(rng.integers(1, k - q + 2)))
        if rng.random() < 0.3:
            os.environ["BMF_ROW_ORDER"] = "far"            # a sample's rows far apart instead of sparsest first
        try:
            f2 = bma.Filter(bma.Params(num_buckets=nb, flags=bma.BMF_FLAG_EARLY_EXIT, **kw))
            f2.load_index(rows, k2i)
        finally:
            for name in ("BMF_PASS1_ROWS", "BMF_MAX_LIVE", "BMF_FOLD", "BMF_FOLD_ROWS", "BMF_ROW_ORDER"):
                os.environ.pop(name, None)
        assert f2.info()["pass1_rows"] >= 1 and f2.info()["pass1_fold"] == (fold if fold else 1)
        # ... in one piece, or in slices whose recounts run on a second stream under the next slice's first pass
        os.environ["BMF_SLICES"] = str(int(rng.choice([1, 3, 8])))
        try:
            out += list(f2.map_windows(bases, quals, ws, wl))
        finally:
            del os.environ["BMF_SLICES"]
        f2.close()
        # ... and the form the library MEASURES to be the fastest on this batch (tune_pruned; threshold lowered to fit)
        os.environ["BMF_TUNE_WINDOWS"] = "64"
        try:
            f3 = bma.Filter(bma.Params(num_buckets=nb, flags=bma.BMF_FLAG_EARLY_EXIT, **kw))
            f3.load_index(rows, k2i)
            c_t, b_t = f3.map_windows(bases, quals, ws, wl)
            assert_same_candidates(c_ref, b_ref, c_t, b_t, "measured choice, the batch that tunes")
            c_t, b_t = f3.map_windows(bases, quals, ws, wl)
            assert_same_candidates(c_ref, b_ref, c_t, b_t, "measured choice, the batch after")
            f3.close()
        finally:
            del os.environ["BMF_TUNE_WINDOWS"]
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("BM_SWEEP_SEEDS", "24"))))   # soak: BM_SWEEP_SEEDS=2000
def test_random_parameters(seed):
    rng = np.random.default_rng(1000 + seed)
    q = int(rng.integers(2, 7))
    k = q + int(rng.integers(0, min(8, 17 - q)))
    S = int(rng.choice([1, 2, 3, 5, 8, 15, 20, 31, 40, 64]))
    F = int(rng.integers(1, min(S, 31) + 1))
    nb = int(rng.choice([1, 31, 64, 127, 128, 129, 1000, 8191, 8193, 20000]))
    max_cand = int(rng.choice([1, 5, 30, 64]))
    density = float(rng.choice([0.02, 0.3, 0.7, 0.97]))
    read_len = int(rng.choice([k, k + 3, 60, 150]))
    threshold = int(rng.choice([0, nb // 2, nb]))
    minq = int(rng.choice([0, 15 * k, 30 * k]))
    c_ref, b_ref, c_got, b_got, c_e, b_e, *two_pass = random_case(
        rng, nb=nb, q=q, k=k, S=S, F=F, max_cand=max_cand, kmer_frac=float(rng.choice([0.3, 1.0])), density=density,
        read_len=read_len, n_reads=160, threshold=threshold, minq=minq, skew=bool(seed % 2))
    what = f"q={q} k={k} S={S} F={F} NB={nb} mc={max_cand} dens={density} L={read_len} thr={threshold} minq={minq}"
    assert_same_candidates(c_ref, b_ref, c_got, b_got, what)
    assert_same_candidates(c_ref, b_ref, c_e, b_e, what + " (early exit)")
    if two_pass:
        assert_same_candidates(c_ref, b_ref, two_pass[0], two_pass[1], what + " (two-pass)")
