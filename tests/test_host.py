"""Host plumbing either side of the filter (no GPU): bucket cutting, the host indexer and its three
file formats, the read simulator, the `bucketmap` command line, the locator and the SAM writer.

The command-line tests drive tests/cpp/bucketmap_oracle: the SAME main.cpp / locator / SAM code as the
product, with the CPU oracle plugged in behind bm::mapper (test infrastructure, never shipped), so the
plumbing is exercised without a GPU.  tests/test_cli_gpu.py repeats the run with the real GPU mapper and
requires an identical SAM file.
"""
import os
import subprocess

import numpy as np
import pytest

from bucket_map_amd import host
from oracle import bm_oracle_np as onp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ORACLE_CLI = os.path.join(ROOT, "tests", "cpp", "bucketmap_oracle")


def build_oracle_cli():
    subprocess.run(["make", "-C", ROOT, "tests/cpp/bucketmap_oracle"], check=True, stdout=subprocess.DEVNULL)
    return ORACLE_CLI


# ------------------------------------------------------------------ FracMinHash (bucket_indexer.h:147-157)

def test_fracminhash_matches_the_reference_header():
    """Row f3's q-gram selection pinned by a reference-owned artefact: tests/golden/fracminhash_ref.json holds what the
    closure of the reference's own hash_function_generator.h (compiled as it lies by make_fracminhash_golden.py)
    selected after std::srand(seed).  select_qgrams with the same (x, y, p) must give the same kmer_to_index."""
    import hashlib
    import json
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "fracminhash_ref.json")))
    inputs = golden["sample_inputs"]
    assert len(golden["cases"]) >= 8
    for c in golden["cases"]:
        k2i = host.select_qgrams_xy(c["q"], c["kmer_frac"], c["x"], c["y"], c["p"], 10000)
        kept = np.flatnonzero(k2i >= 0)
        assert kept.size == c["kept"], c
        assert kept[:24].tolist() == c["first_kept"], c
        assert np.array_equal(k2i[kept], np.arange(kept.size)), "rows are numbered in ascending q-gram"
        assert hashlib.sha256(k2i.tobytes()).hexdigest() == c["k2i_sha256"], c
        # the raw hash values, in Python integers
        assert [(c["x"] * i + c["y"]) % c["p"] % 10000 for i in inputs] == c["sample_hashes"]
        assert c["p"] == 116731                       # first listed prime > 10 * 10000
        assert c["threshold"] == int(np.float32(10000) * np.float32(c["kmer_frac"]))
    # f = 1 keeps everything whatever x, y are -- the setting every BASELINE config runs with
    full = [c for c in golden["cases"] if c["kmer_frac"] == 1.0][0]
    assert full["kept"] == 4 ** full["q"]
    assert np.array_equal(host.select_qgrams(9, 1.0), np.arange(4 ** 9, dtype=np.int32))


# ------------------------------------------------------------------ bucket cutting (utils.h:60-102)

def test_bucket_cutting_rules():
    g = host.Genome.synth(1, [65536 * 2 + 100, 65536 + 301, 500, 300])
    b = g.cut_buckets(65536, 300)
    # record 0: 3 buckets by ceil, the third is 100 bases <= read_len -> skipped
    # record 1: 2 buckets, the second is 301 > 300 -> kept;  record 2: 1 bucket of 500;  record 3: 300 <= 300 skipped
    assert [tuple(x) for x in b] == [(0, 0, 0, 65836), (0, 1, 65536, 131172), (1, 0, 0, 65836), (1, 1, 65536, 65837),
                                     (2, 0, 0, 500)]
    # BM_BUCKET_NUM by the CMake awk rule counts the skipped tails too
    assert g.awk_bucket_num(65536) == 3 + 2 + 1 + 1
    # overlap: every bucket but the last of a record extends read_len bases into the next
    assert b[0][3] - b[0][2] == 65536 + 300


def test_bucket_num_matches_the_reference_script(tmp_path):
    """NB pinned by a reference-owned artefact: tests/golden/bucket_num_ref.json holds what the reference's own
    get_num_buckets.sh (= bucket_map/CMakeLists.txt:13-46) printed for these FASTA files (make_nb_golden.py rebuilds
    them byte for byte).  Held to it: awk_bucket_num on the parsed genome, and the NB the tool announces."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_nb_golden", os.path.join(ROOT, "tests", "golden", "make_nb_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "bucket_num_ref.json")))["cases"]
    assert set(golden) == {c[0] for c in mk.CASES}
    exe = build_oracle_cli()
    for case in mk.CASES:
        name, bucket_len = case[0], case[1]
        path = tmp_path / f"{name}.fa"
        path.write_text(mk.fasta_text(case))
        g = host.Genome.read_fasta(str(path))
        want = golden[name]
        assert want["bucket_len"] == bucket_len
        assert g.awk_bucket_num(bucket_len) == want["bucket_num"], name
        assert [g.record_len(i) for i in range(g.n_records)] == [n for n, _ in case[3]], name
        r = run_cli(exe, ["-x", "-i", name, "--genome", path.name, "--bucket-len", str(bucket_len), "-r", "100", "-k", "5"], tmp_path)
        assert r.returncode == 0, r.stderr
        assert f"number of buckets: {want['bucket_num']}." in r.stderr, (name, r.stderr)
    # the reference's NB can exceed the buckets its indexer keeps (utils.h:88-90 drops tails <= read_len)
    g = host.Genome.read_fasta(str(tmp_path / "dropped_tail.fa"))
    assert len(g.cut_buckets(1024, 100)) == 3 < golden["dropped_tail"]["bucket_num"] == 5


def test_fasta_roundtrip(tmp_path):
    g = host.Genome.synth(2, [1000, 61, 60, 1])
    p = str(tmp_path / "g.fa")
    g.write_fasta(p)
    h = host.Genome.read_fasta(p)
    assert h.n_records == 4
    for i in range(4):
        assert h.record_id(i) == g.record_id(i)
        assert bytes(h.record_seq(i)) == bytes(g.record_seq(i))
    # lower case, U and ambiguity codes: FASTA is read as dna5 then folded to dna4 (N and IUPAC -> A)
    (tmp_path / "x.fa").write_text(">r1 some description\nacgtUNRY\nnnAC\n")
    x = host.Genome.read_fasta(str(tmp_path / "x.fa"))
    assert x.record_id(0) == "r1 some description"
    assert bytes(x.record_seq(0)) == b"ACGTTAAAAAAC"


# ------------------------------------------------------------------ the genome-like generator (bm_synth.h)

def test_genome_like_generator_is_seeded_and_skewed(tmp_path):
    lens = [30_000_000, 9_000_001, 70_000]
    a = host.Genome.synth(77, lens, 1, profile="genome")
    b = host.Genome.synth(77, lens, 5, profile="genome")
    for i in range(len(lens)):                                # thread-count independent, ACGT only
        sa = a.record_seq(i)
        assert a.record_len(i) == lens[i] and np.array_equal(sa, b.record_seq(i))
        assert np.isin(sa, np.frombuffer(b"ACGT", np.uint8)).all()
    c = host.Genome.synth(78, lens, 0, profile="genome")
    assert not np.array_equal(a.record_seq(2), c.record_seq(2))
    # base composition: AT-rich, CpG-depleted, reverse-complement symmetric
    s = a.record_seq(0)
    code = np.zeros(256, np.int64)
    code[list(b"ACGT")] = np.arange(4)
    r = code[s]
    f = np.bincount(r, minlength=4) / len(r)
    assert 0.36 < f[1] + f[2] < 0.50 and abs(f[0] - f[3]) < 0.01 and abs(f[1] - f[2]) < 0.01
    di = np.bincount(r[:-1] * 4 + r[1:], minlength=16) / (len(r) - 1)
    assert di[1 * 4 + 2] < 0.5 * f[1] * f[2]                  # CG
    assert abs(di[0 * 4 + 2] - di[1 * 4 + 3]) < 0.003         # AG ~ CT
    # what the filter sees: the share of index rows that pass the distinguishability threshold is neither 0 nor 1
    # (reference log, GRCh38 at 65 536: 95.8 %, bucketmap_3_map.log:8; uniform bases: 100 % / 0 %)
    for bucket_len, lo, hi in ((65536, 0.90, 0.985), (262144, 0.30, 0.70)):
        nb = a.awk_bucket_num(bucket_len)
        ix = host.Index(a, nb, bucket_len, 300, q=9)
        pop = np.unpackbits(ix.rows(), axis=1).sum(axis=1)
        zeros = np.where(pop == 0, nb, nb - pop)
        share = (zeros >= int(np.float32(0.5) * np.float32(nb))).mean()
        assert lo < share < hi, (bucket_len, share)
    # assembly gaps are runs of A in memory (dna5 -> dna4) and N in the FASTA file; reading it back folds them again
    assert a.gap_bases() > 0
    a.write_fasta(str(tmp_path / "g.fa"))
    text = (tmp_path / "g.fa").read_bytes()
    assert text.count(b"N") == a.gap_bases()
    back = host.Genome.read_fasta(str(tmp_path / "g.fa"))
    assert all(np.array_equal(back.record_seq(i), a.record_seq(i)) for i in range(len(lens)))


# ------------------------------------------------------------------ indexer (bucket_indexer.h:49-127,138-216)

def test_index_against_bruteforce_and_file_formats(tmp_path):
    g = host.Genome.synth(3, [9000, 2500])
    bl, rl, q = 1024, 100, 5
    nb = g.awk_bucket_num(bl) + 2                       # two padding buckets
    ix = host.Index(g, nb, bl, rl, q=q, kmer_frac=1.0, threads=3)
    buckets = g.cut_buckets(bl, rl)
    want = np.zeros((4 ** q, nb), bool)
    for b, (rec, _, s, e) in enumerate(buckets):
        want[onp.kmer_hashes(g.record_seq(rec)[s:e], q), b] = True
    got = onp.unpack_rows(ix.rows(), nb)
    assert np.array_equal(got, want)
    assert np.array_equal(ix.kmer_to_index(), np.arange(4 ** q))
    # files: SURVEY App. B.2
    ix.write(str(tmp_path), "t")
    raw = (tmp_path / "t.qgram").read_bytes()
    assert len(raw) == 4 ** q * ((nb + 7) // 8) and raw == ix.rows().tobytes()
    lines = (tmp_path / "t.kmers_index").read_text().split("\n")
    assert lines[-1] == "" and [int(v) for v in lines[:-1]] == list(range(4 ** q))
    ids = (tmp_path / "t.bucket_id").read_text().split("\n")
    assert ids[:-1] == [g.record_id(int(rec)) for rec, *_ in buckets]


def test_fracminhash_row_selection():
    g = host.Genome.synth(4, [5000])
    ix = host.Index(g, 6, 1024, 100, q=6, kmer_frac=0.25, hash_seed=99)
    k2i = ix.kmer_to_index()
    kept = k2i >= 0
    assert 0.2 < kept.mean() < 0.32                      # threshold 2500 of 10000 hash values, "<="
    assert np.array_equal(k2i[kept], np.arange(kept.sum()))   # rows numbered in ascending q-gram hash
    assert ix.num_rows == kept.sum()
    # same seed -> same selection (the reference seeds from time(); ours is reproducible)
    assert np.array_equal(host.Index(g, 6, 1024, 100, q=6, kmer_frac=0.25, hash_seed=99).kmer_to_index(), k2i)


# ------------------------------------------------------------------ simulator (short_read_simulator.h:157-240)

def test_simulator_is_seeded_and_truthful():
    g = host.Genome.synth(5, [50_000, 20_000])
    a = host.Reads(g, 4096, 150, 150, 500, sub=0, ins=0, dele=0, seed=7, threads=1)
    b = host.Reads(g, 4096, 150, 150, 500, sub=0, ins=0, dele=0, seed=7, threads=4)
    assert np.array_equal(a.bases, b.bases) and np.array_equal(a.truth_bucket, b.truth_bucket)   # thread-count independent
    buckets = g.cut_buckets(4096, 150)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for r in range(200):
        rec, _, s, e = (int(x) for x in buckets[a.truth_bucket[r]])
        o = int(a.truth_offset[r])
        ref = bytes(g.record_seq(rec)[s + o: min(s + o + 150, e)])
        got = bytes(a.bases[int(a.offsets[r]): int(a.offsets[r + 1])])
        assert got == (ref.translate(comp)[::-1] if a.truth_rc[r] else ref)
    assert 0.35 < a.truth_rc.mean() < 0.65
    assert (a.quals == ord("E")).all()
    c = host.Reads(g, 4096, 150, 150, 500, sub=0.02, ins=0.005, dele=0.005, seed=7)
    lens = np.diff(c.offsets.astype(np.int64))
    assert lens.min() < 150 < lens.max()                  # indels change read lengths


# ------------------------------------------------------------------ command line + locator + SAM

@pytest.fixture(scope="module")
def cli_case(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    g = host.Genome.synth(11, [300_000, 120_000, 40_000])
    g.write_fasta(str(d / "g.fa"))
    rd = host.Reads(g, 8192, 150, 150, 1500, seed=5)
    rd.write_fastq(str(d / "reads"))
    return d, g, rd


def run_cli(exe, args, cwd):
    return subprocess.run([exe, *args], cwd=str(cwd), capture_output=True, text=True)


def parse_sam(path):
    header, recs = [], []
    for line in open(path):
        if line.startswith("@"):
            header.append(line.rstrip("\n"))
        else:
            f = line.rstrip("\n").split("\t")
            assert len(f) == 11
            recs.append((f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5], f[9], f[10]))
    return header, recs


def test_cli_index_then_map(cli_case):
    d, g, rd = cli_case
    exe = build_oracle_cli()
    common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1"]
    r = run_cli(exe, ["-x", *common], d)
    assert r.returncode == 0, r.stderr
    for ext in ("qgram", "kmers_index", "bucket_id"):
        assert (d / f"idx.{ext}").exists()
    r = run_cli(exe, [*common, "-q", "reads.fastq", "-o", "out.sam", "--version-check", "0"], d)
    assert r.returncode == 0, r.stderr
    assert "[BENCHMARK]\tElapsed time for bucket mapping:" in r.stderr
    assert "already exists" in r.stderr                   # the indexer refuses to overwrite, then carries on
    header, recs = parse_sam(d / "out.sam")
    assert header[0] == "@HD\tVN:1.6"
    # @SQ: name up to the first space, LN = #buckets * bucket_len (upper bound, bucket_locator.h:491)
    assert header[1:] == ["@SQ\tSN:synth1\tLN:303104", "@SQ\tSN:synth2\tLN:122880", "@SQ\tSN:synth3\tLN:40960"]
    truth = [l.split() for l in open(d / "reads.position_ground_truth")]
    by_read = {}
    for qname, flag, rname, pos, mapq, cigar, seq, qual in recs:
        assert cigar == "*" and flag in (0, 16) and 0 <= mapq <= 60
        by_read.setdefault(int(qname), []).append((flag, rname, pos))
    correct = 0
    for i, t in enumerate(truth):
        ref, pos, rc = int(t[0]), int(t[1]), int(t[2])
        for flag, rname, p in by_read.get(i, []):
            if rname == f"synth{ref + 1}" and abs(p - pos) <= 6 and (flag == 16) == bool(rc):
                correct += 1
                break
    assert correct > 0.97 * len(truth), f"{correct}/{len(truth)} reads at their true position"
    # reads whose exact start is bucket offset 0 are dropped by `offset > 0` -- unmapped
    # reads produce no record at all
    assert len(by_read) <= len(truth)


def test_cli_align_build(cli_case, tmp_path):
    """`bucketmap_align` (main.cpp + -DBM_ALIGN, bucket_locator.h:520-528,560-589) with the CPU oracles behind the
    interfaces: every location verified, MAPQ = 60 + score, CIGAR written, records below -u dropped."""
    import re
    d, g, _ = cli_case
    subprocess.run(["make", "-C", ROOT, "tests/cpp/bucketmap_align_oracle"], check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "cpp", "bucketmap_align_oracle")
    rd = host.Reads(g, 8192, 150, 150, 600, sub=0.02, ins=0.003, dele=0.003, seed=15)
    rd.write_fastq(str(tmp_path / "noisy"))
    common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", "8192", "-r", "150", "-f", "1", "-q", str(tmp_path / "noisy.fastq")]
    r = run_cli(exe, [*common, "-o", str(tmp_path / "a.sam")], d)
    assert r.returncode == 0, r.stderr
    assert "Allowing Smith-Waterman for alignment verifications" in r.stderr
    header, recs = parse_sam(tmp_path / "a.sam")
    assert header[0] == "@HD\tVN:1.6" and len(header) == 4
    truth = [l.split() for l in open(tmp_path / "noisy.position_ground_truth")]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    correct, seen = 0, set()
    for qname, flag, rname, pos, mapq, cigar, seq, qual in recs:
        ops = [(int(n), o) for n, o in re.findall(r"(\d+)([MID])", cigar)]
        assert "".join(f"{n}{o}" for n, o in ops) == cigar and cigar != "*"
        assert sum(n for n, o in ops if o in "MI") == len(seq) == len(qual)
        assert 40 <= mapq <= 60                                        # default -u 40 (main.cpp:36): at most 20 edits
        t = truth[int(qname)]
        ref, tpos, rc = int(t[0]), int(t[1]), int(t[2])
        if flag == 0 and rname == f"synth{ref + 1}" and not rc:
            # forward strand: POS is a genome coordinate; the CIGAR walks the reference and costs 60 - MAPQ edits
            rec = bytes(g.record_seq(ref))
            i, j, cost = 0, pos - 1, 0
            for n, o in ops:
                if o == "M":
                    cost += sum(seq[i + x].encode() != rec[j + x: j + x + 1] for x in range(n))
                    i, j = i + n, j + n
                elif o == "I":
                    cost, i = cost + n, i + n
                else:
                    cost, j = cost + n, j + n
            assert cost == 60 - mapq
            if abs(pos - tpos) <= 6 and qname not in seen:
                correct += 1
                seen.add(qname)
        elif flag == 16 and rname == f"synth{ref + 1}" and rc and qname not in seen:
            correct += 1                                               # (POS on this strand: see bucket_locator.h:576)
            seen.add(qname)
    # (the filter's own sensitivity at 2 % substitutions is ~80 %: reads without a candidate have no record)
    assert len(seen) > 0.7 * len(truth) and correct >= 0.98 * len({x[0] for x in recs}), f"{correct}/{len(truth)}"
    # a stricter -u drops records; MAPQ never falls below it
    r = run_cli(exe, [*common, "-o", str(tmp_path / "b.sam"), "-u", "58"], d)
    assert r.returncode == 0, r.stderr
    _, strict = parse_sam(tmp_path / "b.sam")
    assert 0 < len(strict) < len(recs) and all(x[4] >= 58 for x in strict)


def test_cli_error_behaviour(cli_case):
    d, _, _ = cli_case
    exe = build_oracle_cli()
    r = run_cli(exe, ["--genome", "g.fa"], d)
    assert r.returncode == 255 and "[ERROR]" in r.stderr and "required" in r.stderr      # -i missing -> -1
    r = run_cli(exe, ["-i", "idx", "--genome", "g.fa", "-q", "reads.fastq"], d)
    assert r.returncode == 1 and "output sam file is not set" in r.stderr                # main.cpp:190-193
    r = run_cli(exe, ["-i", "idx", "--genome", "g.fa", "-q", "reads.fastq", "-o", "o2.sam", "-l", "8"], d)
    assert r.returncode == 1 and "query seed length" in r.stderr                          # main.cpp:194-198
    r = run_cli(exe, ["-i", "idx", "--genome", "g.fa", "-q", "reads.txt", "-o", "o2.sam"], d)
    assert r.returncode == 255                                                            # wrong extension
    (d / "exists.sam").write_text("")
    r = run_cli(exe, ["-i", "idx", "--genome", "g.fa", "-q", "reads.fastq", "-o", "exists.sam"], d)
    assert r.returncode == 255 and "already exists" in r.stderr                           # output_file_validator
    r = run_cli(exe, ["-i", "idx", "-q", "reads.fastq", "-o", "o3.sam"], d)
    assert r.returncode == 255 and "BM_BUCKET_NUM" in r.stderr                            # no genome configured
    r = run_cli(exe, ["-i", "idx", "--genome", "g.fa", "--bogus", "1"], d)
    assert r.returncode == 255


# ------------------------------------------------------------------ block readers

def test_block_readers_survive_any_block_size(tmp_path, monkeypatch):
    # the FASTA / FASTQ readers pull fixed-size blocks; with tiny blocks every record straddles a boundary
    g = host.Genome.synth(12, [3000, 1, 777])
    g.write_fasta(str(tmp_path / "g.fa"))
    rd = host.Reads(g, 512, 60, 60, 40, seed=3)
    rd.write_fastq(str(tmp_path / "r"))
    ref = host.fastq_stats(str(tmp_path / "r.fastq"))
    assert ref[0] == 40 and ref[1] == int(rd.offsets[-1])
    for block in (16, 61, 64, 257, 4096):
        monkeypatch.setenv("BM_IO_BLOCK", str(block))
        assert host.fastq_stats(str(tmp_path / "r.fastq")) == ref
        h = host.Genome.read_fasta(str(tmp_path / "g.fa"))
        assert [bytes(h.record_seq(i)) for i in range(3)] == [bytes(g.record_seq(i)) for i in range(3)]
    monkeypatch.delenv("BM_IO_BLOCK")
    # CRLF line ends, blank lines between records, no newline at the end of the file
    text = open(tmp_path / "r.fastq").read().rstrip("\n")
    recs = text.split("\n")
    crlf = "\r\n".join(recs[:8]) + "\r\n\r\n" + "\n".join(recs[8:])
    (tmp_path / "weird.fastq").write_text(crlf)
    assert host.fastq_stats(str(tmp_path / "weird.fastq")) == ref
    (tmp_path / "cut.fastq").write_text("\n".join(recs[:-1]))
    with pytest.raises(RuntimeError):
        host.fastq_stats(str(tmp_path / "cut.fastq"))


def test_fasta_reader_edge_cases(tmp_path):
    """The memory-mapped, multi-threaded FASTA reader: CRLF, blank lines, '>' inside a header, no final newline,
    blanks inside a sequence line, many records (several threads), an empty record, a file without a header."""
    p = tmp_path / "e.fa"
    p.write_bytes(b"\n>r1 with > inside\r\nACGT\r\n\r\nacgu nn\tRY\n>r2\n>r3 empty before\nTTTT\nGG")
    g = host.Genome.read_fasta(str(p))
    assert g.n_records == 3
    assert [g.record_id(i) for i in range(3)] == ["r1 with > inside", "r2", "r3 empty before"]
    assert bytes(g.record_seq(0)) == b"ACGTACGTAAAA" and bytes(g.record_seq(1)) == b"" and bytes(g.record_seq(2)) == b"TTTTGG"
    many = tmp_path / "many.fa"
    rng = np.random.default_rng(3)
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(n))) for n in rng.integers(1, 400, 500)]
    many.write_text("".join(f">s{i}\n" + "\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + "\n" for i, s in enumerate(seqs)))
    g = host.Genome.read_fasta(str(many))
    assert g.n_records == 500 and all(bytes(g.record_seq(i)).decode() == seqs[i] for i in range(500))
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n>late\nAC\n")
    with pytest.raises(Exception):
        host.Genome.read_fasta(str(bad))
    (tmp_path / "empty.fa").write_text("")
    assert host.Genome.read_fasta(str(tmp_path / "empty.fa")).n_records == 0


@pytest.mark.timeout(180)
def test_fastq_index_is_shared_between_passes_and_follows_the_file(tmp_path, monkeypatch):
    """FastqFile (host/bm_genome.h): ONE mapped, progressively built index per file, shared by the tool's passes.  Several
    consumers walk it at once while it is still being built (tiny chunks: dozens of publications) and see the same records;
    a file rewritten under the same name is indexed afresh (size / mtime / inode are part of the key)."""
    from concurrent.futures import ThreadPoolExecutor
    g = host.Genome.synth(77, [90_000])
    host.Reads(g, 4096, 150, 150, 6000, seed=5, noisy_quals=True).write_fastq(str(tmp_path / "a"))
    host.Reads(g, 4096, 150, 150, 2500, seed=6).write_fastq(str(tmp_path / "b"))
    want_a, want_b = host.fastq_stats(str(tmp_path / "a.fastq")), host.fastq_stats(str(tmp_path / "b.fastq"))
    assert want_a[0] == 6000 and want_b[0] == 2500 and want_a != want_b
    monkeypatch.setenv("BM_IO_BLOCK", "4096")
    path = tmp_path / "shared.fastq"
    for src, want in (("a.fastq", want_a), ("b.fastq", want_b), ("a.fastq", want_a)):
        path.write_bytes((tmp_path / src).read_bytes())               # same name, new content
        with ThreadPoolExecutor(6) as pool:
            got = list(pool.map(lambda _: host.fastq_stats(str(path)), range(6)))
        assert got == [want] * 6
    # an error in the middle of the file: every consumer gets the records before it, then the error
    text = path.read_text().split("\n")
    text[4 * 3000 + 2] = "-"                                             # record 3000 loses its '+' line
    path.write_text("\n".join(text))
    with ThreadPoolExecutor(4) as pool:
        def attempt(_):
            try:
                host.fastq_stats(str(path))
                return "no error"
            except RuntimeError as e:
                return str(e)
        errs = list(pool.map(attempt, range(4)))
    assert all("FASTQ" in e and "shared.fastq" in e for e in errs), errs


def test_fastq_reader_finds_record_starts_anywhere(tmp_path, monkeypatch):
    """The mapped FASTQ reader resynchronises at every chunk boundary on "a line that begins with '@' whose line after
    next begins with '+'".  Qualities that begin with '@' or '+', headers that look like anything, reads of every
    length (empty ones included) and chunk sizes from 16 bytes up must all give the records a plain line-by-line
    parse gives; damage anywhere must raise."""
    rng = np.random.default_rng(17)
    recs = []
    for i in range(300):
        n = int(rng.integers(0, 90))
        seq = "".join("ACGTN"[j] for j in rng.integers(0, 5, n))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 127, n))
        if n and i % 3 == 0:
            qual = "@" + qual[1:]                      # a quality line that looks like a header
        if n and i % 5 == 0:
            qual = "+" + qual[1:]                      # ... or like the separator
        recs.append((f"r{i} @+ {'@' * (i % 4)}", seq, qual))
    text = "".join(f"@{i}\n{s}\n+\n{q}\n" for i, s, q in recs)
    path = tmp_path / "adv.fastq"
    path.write_text(text)

    def stats(p):
        return host.fastq_stats(str(p))

    def fnv(parts):
        h = 1469598103934665603
        for v in parts:
            for c in v.encode():
                h = ((h ^ c) * 1099511628211) & (2 ** 64 - 1)
            h = ((h ^ 0xFF) * 1099511628211) & (2 ** 64 - 1)
        return h
    want = (len(recs), sum(len(s) for _, s, _ in recs), fnv([x for r in recs for x in r]))
    for block in (16, 17, 23, 64, 100, 333, 1024, 1 << 20):
        monkeypatch.setenv("BM_IO_BLOCK", str(block))
        assert stats(path) == want, block
    # a last record with an empty sequence and no trailing newline ends at its '+' line: accepted by every path
    (tmp_path / "tail.fastq").write_text(text + "@last\n\n+\n")
    for block in (16, 64, 1 << 20):
        monkeypatch.setenv("BM_IO_BLOCK", str(block))
        assert stats(tmp_path / "tail.fastq")[:2] == (want[0] + 1, want[1]), block
    # a third line that does not begin with '+' is refused wherever the record lies: in the first chunk, in a later one
    # (found by find_record) and by the block reader that serves pipes
    noplus = text.replace("\n+\n", "\n-\n", 1)
    (tmp_path / "noplus.fastq").write_text(noplus)
    (tmp_path / "noplus_late.fastq").write_text(text + "@x\nACGT\n-\nIIII\n")
    for name in ("noplus.fastq", "noplus_late.fastq"):
        for block in (16, 64, 1 << 20):
            monkeypatch.setenv("BM_IO_BLOCK", str(block))
            with pytest.raises(RuntimeError):
                stats(tmp_path / name)
    fifo = tmp_path / "pipe.fastq"
    os.mkfifo(fifo)
    import threading
    for body, ok in ((text + "@last\n\n+\n", True), (noplus, False)):
        t = threading.Thread(target=lambda b=body: open(fifo, "w").write(b))
        t.start()
        try:
            if ok:
                assert stats(fifo)[:2] == (want[0] + 1, want[1])
            else:
                with pytest.raises(RuntimeError):
                    stats(fifo)
        finally:
            t.join()
    # damage: a missing quality line in the middle, junk between two records, unequal lengths
    lines = text.split("\n")
    for what, bad in (("missing line", lines[:403] + lines[404:]), ("junk", lines[:400] + ["junk"] + lines[400:]),
                      ("lengths", lines[:401] + [lines[401] + "A"] + lines[402:])):
        (tmp_path / "bad.fastq").write_text("\n".join(bad))
        for block in (16, 64, 1 << 20):
            monkeypatch.setenv("BM_IO_BLOCK", str(block))
            with pytest.raises(RuntimeError):
                stats(tmp_path / "bad.fastq")
