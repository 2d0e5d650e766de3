"""pytest configuration: the `gpu` marker, import paths, in-tree builds and shared fixtures.

`-m "not gpu"` runs the oracle, host-logic and ABI-surface tests (no GPU needed); `-m gpu` runs the
parity tests proper, calling the HIP path through the C ABI and comparing with the oracle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build in-tree libraries if they are missing (the GPU box receives them prebuilt).
    need = [os.path.join(ROOT, "bucket-map_amd", "libbmf.so"), os.path.join(ROOT, "bucket-map_amd", "libbmhost.so"),
            os.path.join(ROOT, "oracle", "libbm_oracle.so"), os.path.join(ROOT, "bucket-map_amd", "bucketmap"),
            os.path.join(ROOT, "tests", "cpp", "bucketmap_oracle"), os.path.join(ROOT, "bucket-map_amd", "bucketmap_align"),
            os.path.join(ROOT, "tests", "cpp", "bucketmap_align_oracle")]
    # In the build container (no GPU) always let make decide, so that the libraries the GPU box receives are never
    # older than the sources; on the GPU box only build what is missing.
    if not os.path.exists("/dev/kfd") or not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-C", ROOT, "-j4"], check=True, stdout=subprocess.DEVNULL)


def have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Case:
    """A small synthetic genome + index + simulated reads, shared by oracle and GPU tests."""

    def __init__(self, *, record_lengths, bucket_len, read_len, n_reads, q=9, k=12, samples=15, error_rate=0.4,
                 distinguishability=0.5, base_quality=25, kmer_frac=1.0, sub=0.002, ins=0.00025, dele=0.00025,
                 noisy_quals=False, seed=20240001, extra_buckets=0, sim_read_len=None, profile="uniform"):
        from bucket_map_amd import host
        self.genome = host.Genome.synth(seed, record_lengths, profile=profile)
        self.bucket_len, self.read_len = bucket_len, read_len
        # NB as the reference's CMake awk rule gives it (can exceed the kept buckets) + optional padding
        self.num_buckets = self.genome.awk_bucket_num(bucket_len) + extra_buckets
        self.index = host.Index(self.genome, self.num_buckets, bucket_len, read_len, q=q, kmer_frac=kmer_frac)
        self.reads = host.Reads(self.genome, bucket_len, read_len, sim_read_len or read_len, n_reads, sub=sub, ins=ins,
                                dele=dele, seed=seed + 2, noisy_quals=noisy_quals)
        self.cli = dict(index_seed=q, query_seed=k, read_len=read_len, mapper_samples=samples,
                        max_error_rate=error_rate, distinguishability=distinguishability,
                        average_base_quality=base_quality)

    def oracle_index(self):
        from oracle import oracle_c
        p = oracle_c.params_from_cli(self.num_buckets, **self.cli)
        return oracle_c.Index(p, rows_ptr=self.index.rows_ptr, n_rows=self.index.num_rows,
                              k2i_ptr=self.index.k2i_ptr, n_kmers=self.index.num_kmers)

    def gpu_filter(self, flags=0):
        import bucket_map_amd as bma
        f = bma.Filter(bma.Params.from_cli(self.num_buckets, flags=flags, **self.cli))
        f.load_index_ptr(self.index.rows_ptr, self.index.num_rows, self.index.k2i_ptr, self.index.num_kmers)
        return f


def oracle_map_windows(ix, bases, quals, win_start, win_len, threads=None):
    """ix.map_windows over `threads` host threads (the oracle is re-entrant, ctypes releases the GIL): the way the
    CPU restatement reaches 100 k-read samples inside a GPU test.  Returns (counts, buckets, rows ANDed)."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(win_start)
    threads = threads or max(1, min(len(os.sched_getaffinity(0)), 32))
    if n < 4 * threads:
        return ix.map_windows(bases, quals, win_start, win_len)
    cuts = [n * t // threads for t in range(threads + 1)]
    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(lambda t: ix.map_windows(bases, quals, win_start[cuts[t]:cuts[t + 1]], win_len[cuts[t]:cuts[t + 1]]),
                              range(threads)))
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), sum(int(p[2]) for p in parts))


def assert_same_candidates(c_ref, b_ref, c_got, b_got, what=""):
    """Bit-exact comparison of candidate lists: counts, then ids (values AND order)."""
    c_ref, c_got = np.asarray(c_ref), np.asarray(c_got)
    assert c_ref.shape == c_got.shape
    bad = np.nonzero((c_ref != c_got).any(axis=1))[0]
    assert bad.size == 0, f"{what}: counts differ at windows {bad[:10]}: ref {c_ref[bad[:3]]} got {c_got[bad[:3]]}"
    mc = b_ref.shape[-1]
    mask = np.arange(mc)[None, None, :] < c_ref[:, :, None]
    diff = (b_ref != b_got) & mask
    badw = np.nonzero(diff.any(axis=(1, 2)))[0]
    assert badw.size == 0, f"{what}: bucket ids differ at windows {badw[:10]}"


@pytest.fixture(scope="session")
def ecoli_like():
    """Config 1 shape, shrunk: one record, NB crossing a 64-bit word boundary with padding bits."""
    return Case(record_lengths=[150_000 + 4_641_652 % 65536], bucket_len=2048, read_len=150, n_reads=400,
                extra_buckets=1)
