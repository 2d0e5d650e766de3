"""End to end on the GPU: the `bucketmap` tool (GPU mapper behind bm::mapper) must write the SAME SAM
file as the same tool with the CPU oracle behind bm::mapper -- candidate buckets feed the locator, so
identical SAM means identical candidate sets AND identical mapping positions (north_star).  Also
covers the multi-context split of one batch (the code path used with several GPUs) and full-size
properties of the filter at the BASELINE geometry."""
import os
import subprocess

import numpy as np
import pytest

from conftest import Case, assert_same_candidates, oracle_map_windows

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GPU_CLI = os.path.join(ROOT, "bucket-map_amd", "bucketmap")
ORACLE_CLI = os.path.join(ROOT, "tests", "cpp", "bucketmap_oracle")


def _run(exe, args, cwd, env=None):
    r = subprocess.run([exe, *args], cwd=str(cwd), capture_output=True, text=True,
                       env=None if env is None else {**os.environ, **env})
    assert r.returncode == 0, r.stderr
    return r.stderr


@pytest.mark.parametrize("gpus,long_reads", [("0", False), ("0,0", False), ("0,0,0", True)])
def test_sam_identical_to_oracle_backed_run(tmp_path, gpus, long_reads):
    from bucket_map_amd import host
    g = host.Genome.synth(21, [400_000, 150_000, 30_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    rd = host.Reads(g, 8192, 150, 900 if long_reads else 150, 400 if long_reads else 3000, sub=0.01, seed=6)
    rd.write_fastq(str(tmp_path / "reads"))
    common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1", "-q", "reads.fastq"]
    _run(GPU_CLI, ["-x", "-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1"], tmp_path)
    err = _run(GPU_CLI, [*common, "-o", "gpu.sam", "--gpus", gpus], tmp_path)
    assert "Elapsed time for bucket mapping" in err
    _run(ORACLE_CLI, [*common, "-o", "cpu.sam"], tmp_path)
    gpu_sam, cpu_sam = (tmp_path / "gpu.sam").read_bytes(), (tmp_path / "cpu.sam").read_bytes()
    assert gpu_sam == cpu_sam
    assert gpu_sam.count(b"\n") > 0.9 * rd.n


def test_many_small_batches_keep_the_order(tmp_path):
    # 3000 reads in batches of 257: the two-slot pipeline (parse batch i+1 while the devices work on batch i)
    # must scatter results in batch order -- the SAM file is order sensitive through the locator
    from bucket_map_amd import host
    g = host.Genome.synth(25, [350_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    host.Reads(g, 8192, 150, 150, 3000, sub=0.01, seed=11).write_fastq(str(tmp_path / "reads"))
    common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1", "-q", "reads.fastq"]
    _run(GPU_CLI, [*common, "-o", "small.sam", "--gpus", "0,0"], tmp_path, env={"BM_BATCH_READS": "257"})
    _run(GPU_CLI, [*common, "-o", "one.sam", "--no-early-exit"], tmp_path)     # the filter without exact pruning
    _run(ORACLE_CLI, [*common, "-o", "cpu.sam"], tmp_path, env={"BM_BATCH_READS": "1000"})
    ref = (tmp_path / "cpu.sam").read_bytes()
    assert (tmp_path / "small.sam").read_bytes() == ref and (tmp_path / "one.sam").read_bytes() == ref


def test_pass_order_and_batching_do_not_change_the_sam_file(tmp_path):
    # the locator's passes beside map() (default) or after it (BM_SERIAL_PASSES=1, the measurement knob), the ramped batches or
    # fixed ones, one gather thread or many, tiny index chunks: byte-identical SAM
    from bucket_map_amd import host
    g = host.Genome.synth(29, [500_000, 40_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    host.Reads(g, 8192, 150, 150, 40_000, sub=0.01, seed=13).write_fastq(str(tmp_path / "reads"))
    common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1", "-q", "reads.fastq"]
    _run(GPU_CLI, [*common, "-o", "default.sam"], tmp_path)
    _run(GPU_CLI, [*common, "-o", "serial.sam"], tmp_path, env={"BM_SERIAL_PASSES": "1"})
    _run(GPU_CLI, [*common, "-o", "odd.sam", "--gpus", "0,0"], tmp_path,
         env={"BM_BATCH_READS": "7001", "BMF_GATHER_THREADS": "5", "BMF_PIECE_WINDOWS": "1111", "BM_IO_BLOCK": "65536"})
    ref = (tmp_path / "default.sam").read_bytes()
    assert ref.count(b"\n") > 0.9 * 40_000
    assert (tmp_path / "serial.sam").read_bytes() == ref and (tmp_path / "odd.sam").read_bytes() == ref


def test_long_read_profile_sam_identical(tmp_path):
    # the reference's long-read command line (benchmark/long_read/benchmark_map.sh:25):
    # -s 30 -e 0.9 -n 0.1 -l 12 -p 20 -u 5 on ONT-like reads (sub 0.03, ins = del 0.025), bucket_len 262144
    from bucket_map_amd import host
    g = host.Genome.synth(23, [2_400_000, 700_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    rd = host.Reads(g, 262144, 300, 6000, 150, sub=0.03, ins=0.025, dele=0.025, seed=8)
    rd.write_fastq(str(tmp_path / "reads"))
    flags = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "262144", "-f", "1", "-s", "30", "-e", "0.9", "-n", "0.1",
             "-l", "12", "-p", "20", "-u", "5", "-q", "reads.fastq"]
    _run(GPU_CLI, [*flags, "-o", "gpu.sam", "--gpus", "0,0"], tmp_path)
    _run(ORACLE_CLI, [*flags, "-o", "cpu.sam"], tmp_path)
    gpu_sam = (tmp_path / "gpu.sam").read_bytes()
    assert gpu_sam == (tmp_path / "cpu.sam").read_bytes()
    # most long reads get at least one location despite 8 % errors
    names = {line.split(b"\t", 1)[0] for line in gpu_sam.split(b"\n") if line and not line.startswith(b"@")}
    assert len(names) > 0.5 * rd.n


@pytest.mark.parametrize("frac", ["1", "0.25"])
def test_gpu_index_writes_identical_files(tmp_path, frac):
    # bucketmap -x --gpu-index must write byte-identical .qgram / .kmers_index / .bucket_id
    from bucket_map_amd import host
    g = host.Genome.synth(24, [500_000, 90_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    common = ["--genome", "g.fa", "--bucket-len", "4096", "-r", "150", "-f", frac]
    _run(GPU_CLI, ["-x", "-i", "host", "--host-index", *common], tmp_path)
    err = _run(GPU_CLI, ["-x", "-i", "gpu", "--gpu-index", *common], tmp_path)
    assert "stored in" in err
    for ext in ("qgram", "kmers_index", "bucket_id"):
        assert (tmp_path / f"gpu.{ext}").read_bytes() == (tmp_path / f"host.{ext}").read_bytes(), ext


def test_gpu_index_is_not_bound_by_the_mappers_limits(tmp_path):
    # index-only runs must not inherit the mapper's limits: -r beyond 16 384 (the filter's longest window) and a
    # query seed far from the index seed (k - q + 1 > 8) still build on the device, byte-identical to the host indexer
    from bucket_map_amd import host
    g = host.Genome.synth(28, [300_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    common = ["--genome", "g.fa", "--bucket-len", "65536", "-r", "20000", "-k", "4", "-l", "14", "-f", "1"]
    _run(GPU_CLI, ["-x", "-i", "host", "--host-index", *common], tmp_path)
    _run(GPU_CLI, ["-x", "-i", "gpu", "--gpu-index", *common], tmp_path)
    _run(GPU_CLI, ["-x", "-i", "auto", *common], tmp_path)
    for ext in ("qgram", "kmers_index", "bucket_id"):
        ref = (tmp_path / f"host.{ext}").read_bytes()
        assert (tmp_path / f"gpu.{ext}").read_bytes() == ref and (tmp_path / f"auto.{ext}").read_bytes() == ref, ext
    # q = 12 is beyond the device build (3 <= q <= 10): the default falls back to the host indexer instead of failing
    _run(GPU_CLI, ["-x", "-i", "q12", "--genome", "g.fa", "--bucket-len", "65536", "-r", "150", "-k", "12", "-l", "14", "-f", "1"], tmp_path)
    assert (tmp_path / "q12.qgram").stat().st_size == 4 ** 12 * 1


def test_cli_without_index_files_builds_them(tmp_path):
    # locator::initialize indexes first when the files are missing (locator.h:33-34)
    from bucket_map_amd import host
    g = host.Genome.synth(22, [100_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    host.Reads(g, 8192, 150, 150, 200, seed=7).write_fastq(str(tmp_path / "reads"))
    _run(GPU_CLI, ["-i", "fresh", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1", "-q", "reads.fastq",
                   "-o", "o.sam"], tmp_path)
    assert (tmp_path / "fresh.qgram").exists() and (tmp_path / "o.sam").stat().st_size > 0


def test_full_size_properties():
    """BASELINE configs[1] geometry (Egu-like 1.70 Gbp, NB ~ 26.4 k, -f 1 index, 300 bp reads) with a
    reduced read count: oracle parity on a sample, source bucket recovered, idempotence, permutation
    and batch-split invariance -- properties that do not depend on the batch size."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    import bucket_map_amd as bma
    from bucket_map_amd import host
    from oracle import oracle_c

    genome = host.Genome.synth(20240001, bench.egu_like_record_lengths(1_701_312_507))
    nb = genome.awk_bucket_num(65536)
    assert 26_000 < nb < 27_000
    index = host.Index(genome, nb, 65536, 300, q=9)
    reads = host.Reads(genome, 65536, 300, 300, 200_000, seed=20240003)
    cli = dict(read_len=300)
    flt = bma.Filter(bma.Params.from_cli(nb, **cli))
    flt.load_index_ptr(index.rows_ptr, index.num_rows, index.k2i_ptr, index.num_kmers)
    # (0) the GPU index build gives the host indexer's rows byte for byte at full size too
    fb = bma.Filter(bma.Params.from_cli(nb, **cli))
    flat, _ = genome.flat()
    bstart, blen = genome.bucket_views(65536, 300)
    fb.build_index(flat, bstart, blen, index.kmer_to_index())
    del flat
    assert np.array_equal(fb.index_download(), index.rows())
    fb.close()
    ws, wl, _, _ = bma.windows_for_reads(reads.offsets, 300)
    c, b = flt.map_windows(reads.bases, reads.quals, ws, wl)
    # (a) parity with the oracle on 100 000 reads (SURVEY 7 step 4's gate; the CPU restatement on all host cores)
    n = 100_000
    ora = oracle_c.Index(oracle_c.params_from_cli(nb, **cli), rows_ptr=index.rows_ptr, n_rows=index.num_rows,
                         k2i_ptr=index.k2i_ptr, n_kmers=index.num_kmers)
    c_ref, b_ref, _ = oracle_map_windows(ora, reads.bases, reads.quals, ws[:n], wl[:n])
    assert_same_candidates(c_ref, b_ref, c[:n], b[:n], "full-size sample")
    # ... the same 100 000 through the pruning kernels the tools use by default
    fp = bma.Filter(bma.Params.from_cli(nb, flags=bma.BMF_FLAG_EARLY_EXIT, **cli))
    fp.load_index_ptr(index.rows_ptr, index.num_rows, index.k2i_ptr, index.num_kmers)
    assert fp.info()["pass1_rows"] >= 1
    c_p, b_p = fp.map_windows(reads.bases, reads.quals, ws[:n], wl[:n])
    assert_same_candidates(c_ref, b_ref, c_p, b_p, "full-size sample, two-pass pruning")
    fp.close()
    # (b) reads recover their source bucket on their strand, as the reference's logs report (95-99 %)
    s = reads.truth_rc.astype(np.int64)
    i = np.arange(reads.n)
    own = b[i, s]
    valid = np.arange(own.shape[1])[None, :] < c[i, s][:, None]
    assert ((own == reads.truth_bucket[:, None]) & valid).any(axis=1).mean() > 0.97
    # (c) lists are ascending and within range
    for o in (0, 1):
        lst = b[:, o, :].astype(np.int64)
        m = np.arange(lst.shape[1])[None, :] < c[:, o][:, None]
        assert (lst[m] < nb).all()
        asc = (np.diff(lst, axis=1) > 0) | ~m[:, 1:]
        assert asc.all()
    # (d) idempotence, permutation, batch split
    c2, b2 = flt.map_windows(reads.bases, reads.quals, ws, wl)
    assert_same_candidates(c, b, c2, b2, "idempotence")
    perm = np.random.default_rng(3).permutation(len(ws))
    cp, bp = flt.map_windows(reads.bases, reads.quals, ws[perm], wl[perm])
    assert_same_candidates(c[perm], b[perm], cp, bp, "permutation")
    h = 77_777
    ca, ba = flt.map_windows(reads.bases, reads.quals, ws[:h], wl[:h])
    assert_same_candidates(c[:h], b[:h], ca, ba, "batch split")
    flt.close()


def test_sam_identical_with_fracminhash_index(tmp_path):
    # the reference's default -f 0.25: three q-grams in four are not indexed (seeded selection: both tools keep the same rows)
    from bucket_map_amd import host
    g = host.Genome.synth(26, [500_000, 60_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    rd = host.Reads(g, 8192, 150, 150, 2500, sub=0.005, seed=12)
    rd.write_fastq(str(tmp_path / "reads"))
    common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "0.25", "--hash-seed", "7", "-q", "reads.fastq"]
    _run(GPU_CLI, [*common, "-o", "gpu.sam"], tmp_path)
    _run(ORACLE_CLI, [*common, "-o", "cpu.sam"], tmp_path)
    gpu_sam = (tmp_path / "gpu.sam").read_bytes()
    assert gpu_sam == (tmp_path / "cpu.sam").read_bytes()
    assert gpu_sam.count(b"\n") > 0.8 * rd.n


def test_mapper_test_tool_and_distinguishability_line(tmp_path):
    """The reference's benchmark-only entry points (_query_file / _check_ground_truth, q_gram_mapper.h:560-636, driven by
    mapper_test.cpp) and distinguishability_filter::read's log line (:183-185), on the GPU filter."""
    import re
    from bucket_map_amd import host
    g = host.Genome.synth(27, [600_000, 70_000])
    g.write_fasta(str(tmp_path / "g.fa"))
    rd = host.Reads(g, 8192, 150, 150, 4000, sub=0.005, seed=13)
    rd.write_fastq(str(tmp_path / "reads"))
    common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1"]
    _run(GPU_CLI, ["-x", *common], tmp_path)
    # _query_file hands WHOLE records to query_sequence (no truncation to -r): reads with insertions are 151+ bases,
    # so the mapper is built for -r 200 (the index files do not depend on the mapper's -r)
    err = _run(os.path.join(ROOT, "bucket-map_amd", "mapper_test"),
               ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "200", "-f", "1", "-q", "reads.fastq",
                "--ground-truth", "reads.bucket_ground_truth"], tmp_path)
    m = re.search(r"Number of Q-grams with distinguishability >= ([0-9.]+): (\d+) \(", err)
    assert m and abs(float(m.group(1)) - 0.5) < 0.01
    # every row of this index has more zeros than half the buckets except the densest few: count them on the host
    index = host.Index(g, g.awk_bucket_num(8192), 8192, 150, q=9)
    nb = g.awk_bucket_num(8192)
    ones = np.unpackbits(index.rows(), axis=1, bitorder="little")[:, :nb].sum(axis=1)
    thr = int(np.float32(0.5) * np.float32(nb))
    assert int(m.group(2)) == int((nb - ones > thr).sum())
    assert f"Total number of sequences: {rd.n}." in err
    correct = int(re.search(r"Correct bucket predictions: (\d+) ", err).group(1))
    assert correct > 0.97 * rd.n
    for line in ("Elapsed time for bucket query", "Average number of buckets returned", "no candidate bucket",
                 "uniquely mapped sequences", "mapped to <= 5 buckets", "mapped to <= 10 buckets"):
        assert line in err
    # the same numbers from the C ABI directly: source bucket among the candidates of the true strand
    import bucket_map_amd as bma
    flt = bma.Filter(bma.Params.from_cli(nb, read_len=200))
    flt.load_index_ptr(index.rows_ptr, index.num_rows, index.k2i_ptr, index.num_kmers)
    ws, wl, _, _ = bma.windows_for_reads(rd.offsets, 200)
    c, b = flt.map_windows(rd.bases, rd.quals, ws, wl)
    flt.close()
    s = rd.truth_rc.astype(np.int64)
    i = np.arange(rd.n)
    own, valid = b[i, s], np.arange(b.shape[2])[None, :] < c[i, s][:, None]
    assert correct == int(((own == rd.truth_bucket[:, None]) & valid).any(axis=1).sum())
