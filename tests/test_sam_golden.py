"""The tools against tests/golden/sam_small.json -- expected @SQ lines and SAM records worked out by a
plain-Python statement of the whole reference tool (tests/golden/py_bucketmap.py, written from
bucket_map/locator/bucket_locator.h:209-290,292-405,455-705 and q_gram_mapper.h:380-557) that shares no code
with bucket-map_amd/host/, oracle/*.c or the kernels.

This is the independent check of the bucket loop's ordering contract (:651-693), _filter_best_locations
(:350-405), the .bucket_id -> @SQ collapse (:473-503) and the record fields (:544-600): the oracle-backed
tool compiles the same host headers as the product, so comparing those two compares that code with itself.

CPU: the oracle-backed tools (C oracles behind bm::mapper / offset_scanner / alignment_verifier).
GPU: the product tools, one device and the --gpus 0,0,0 split.
"""
import json
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TOOLS = {
    ("bucketmap", "oracle"): os.path.join(ROOT, "tests", "cpp", "bucketmap_oracle"),
    ("bucketmap_align", "oracle"): os.path.join(ROOT, "tests", "cpp", "bucketmap_align_oracle"),
    ("bucketmap", "gpu"): os.path.join(ROOT, "bucket-map_amd", "bucketmap"),
    ("bucketmap_align", "gpu"): os.path.join(ROOT, "bucket-map_amd", "bucketmap_align"),
}


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "sam_small.json")) as f:
        return json.load(f)


def write_inputs(golden, d):
    with open(d / "g.fa", "w") as f:
        for name, seq in golden["records"]:
            f.write(f">{name}\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
    with open(d / "reads.fastq", "w") as f:
        for name, seq, qual in golden["reads"]:
            f.write(f"@{name}\n{seq}\n+\n{qual}\n")


def run_tool(golden, which, backend, d, extra=(), env=None):
    fl = golden["flags"]
    args = ["-i", "idx", "--genome", "g.fa", "--bucket-len", str(fl["bucket_len"]), "-r", str(fl["read_len"]),
            "-k", str(fl["q"]), "-l", str(fl["k"]), "-s", str(fl["S"]), "-e", str(fl["e"]), "-d", str(fl["d"]),
            "-b", str(fl["b"]), "-n", str(fl["n"]), "-p", str(fl["p"]), "-u", str(fl["u"]), "-f", "1",
            "-q", "reads.fastq", "-o", f"{which}_{backend}.sam", *extra]
    r = subprocess.run([TOOLS[(which, backend)], *args], cwd=str(d), capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    sq, recs = [], []
    for line in open(d / f"{which}_{backend}.sam"):
        f = line.rstrip("\n").split("\t")
        if f[0] == "@SQ":
            sq.append([f[1][3:], int(f[2][3:])])
        elif not line.startswith("@"):
            assert len(f) == 11 and f[6:9] == ["*", "0", "0"]
            recs.append([f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5], f[9], f[10]])
    return sq, recs, r.stderr


def check(golden, which, sq, recs, stderr):
    exp = golden[which]
    assert f"number of buckets: {exp['num_buckets']}." in stderr          # NB by the CMake awk rule
    assert sq == exp["sq"]
    # record by record, in file order: QNAME, FLAG, RNAME, POS, MAPQ, CIGAR, SEQ, QUAL
    for got, want in zip(recs, exp["sam"]):
        assert got == want
    assert len(recs) == len(exp["sam"])


def test_fixture_covers_the_order_sensitive_cases(golden):
    """The fixture is only worth something if the cases it was built for actually occur in it."""
    recs = golden["bucketmap"]["sam"]
    by_name = {}
    for r in recs:
        by_name.setdefault(r[0], []).append(r)
    names = [r[0] for r in golden["reads"]]
    assert golden["bucketmap"]["num_buckets"] == golden["bucketmap"]["kept_buckets"] + 1     # a dropped tail bucket
    assert [s[0] for s in golden["bucketmap"]["sq"]] == ["chrA", "chrB", "chrC"]             # two records, one @SQ
    assert by_name["part2"][0][2] == "chrA" and by_name["part2"][0][3] == 6 * 256 + 300 + 1  # offsets carried across
    assert "record_start" not in by_name and "record_start_rc" not in by_name               # offset 0 is dropped
    assert by_name["bucket_edge"][0][3] == 257                                               # found through bucket 0
    assert len(by_name["dup"]) == 2 and {r[2] for r in by_name["dup"]} == {"chrB", "chrC"}   # vote tie, key order
    assert len(by_name["dup_partial"]) == 1
    assert any(r[1] == 16 for r in recs) and any(r[1] == 0 for r in recs)
    assert by_name["long200"][0][4] == 60 and len(by_name["long200"]) == 1                   # five windows merged
    assert len({r[4] for r in recs}) >= 5                                                    # MAPQ = min(60, 6 votes)
    assert by_name["with_N"][0][6].count("N") == 0 and by_name["lower_case"][0][6].isupper()  # SEQ folded to dna4
    for gone in ("low_quality", "short5", "junk0"):
        assert gone in names and gone not in by_name
    al = golden["bucketmap_align"]["sam"]
    assert any("D" in r[5] for r in al) and any("I" in r[5] for r in al)
    assert len([r for r in al if r[0] == "long200"]) == 5                                    # no merge with BM_ALIGN
    assert any(r[4] > 120 for r in al)                                                       # 60u + score wrapped


@pytest.mark.parametrize("which", ["bucketmap", "bucketmap_align"])
def test_oracle_backed_tool_equals_python_fixture(golden, tmp_path, which):
    write_inputs(golden, tmp_path)
    check(golden, which, *run_tool(golden, which, "oracle", tmp_path))


@pytest.mark.parametrize("block_reads", ["1", "3", "7"])
def test_verified_records_keep_their_order_across_blocks(golden, tmp_path, block_reads):
    """bucketmap_align verifies a block of reads while it writes the records of the block before and reads the next one:
    with blocks of 1, 3 and 7 reads the file is still the fixture's, record for record (batches of a few reads for the
    mapper too)."""
    write_inputs(golden, tmp_path)
    check(golden, "bucketmap_align", *run_tool(golden, "bucketmap_align", "oracle", tmp_path,
                                              env={"BM_VERIFY_BLOCK_READS": block_reads, "BM_BATCH_READS": "5"}))


@pytest.mark.gpu
def test_gpu_verified_records_keep_their_order_across_blocks(golden, tmp_path):
    write_inputs(golden, tmp_path)
    check(golden, "bucketmap_align", *run_tool(golden, "bucketmap_align", "gpu", tmp_path, extra=["--gpus", "0,0"],
                                              env={"BM_VERIFY_BLOCK_READS": "4", "BM_BATCH_READS": "5"}))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["bucketmap", "bucketmap_align"])
@pytest.mark.parametrize("gpus", ["0", "0,0,0"])
def test_gpu_tool_equals_python_fixture(golden, tmp_path, which, gpus):
    write_inputs(golden, tmp_path)
    check(golden, which, *run_tool(golden, which, "gpu", tmp_path, extra=["--gpus", gpus]))
    # and without exact pruning / with the host indexer: same file
    if gpus == "0":
        os.remove(tmp_path / f"{which}_gpu.sam")
        check(golden, which, *run_tool(golden, which, "gpu", tmp_path, extra=["--no-early-exit"]))
