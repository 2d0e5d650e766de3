"""The host plumbing and the C oracles under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5's race /
memory-error detection, CPU build: GPU sanitizers are not available on the pool).

`make tests/cpp/bucketmap[_align]_oracle_asan` compiles main.cpp, the locator, the SAM writer, the FASTA/FASTQ readers
and the three C oracles with -fsanitize=address,undefined; the tools then index and map the inputs of
tests/golden/sam_small.json -- short reads, a 5-window read, N / IUPAC / lower case, low qualities, both strands -- and must
(i) exit 0 with no sanitizer report and (ii) still write the fixture's records.
"""
import json
import os
import subprocess

import pytest

from test_sam_golden import ROOT, check, write_inputs

ASAN = {"bucketmap": os.path.join(ROOT, "tests", "cpp", "bucketmap_oracle_asan"),
        "bucketmap_align": os.path.join(ROOT, "tests", "cpp", "bucketmap_align_oracle_asan")}


@pytest.fixture(scope="module")
def asan_tools():
    r = subprocess.run(["make", "-C", ROOT, "-j2", *[os.path.relpath(p, ROOT) for p in ASAN.values()]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return ASAN


@pytest.mark.parametrize("which", ["bucketmap", "bucketmap_align"])
@pytest.mark.parametrize("blocks", [None, {"BM_VERIFY_BLOCK_READS": "3", "BM_BATCH_READS": "5", "BM_IO_BLOCK": "64"}])
def test_sanitized_tool_is_clean_and_writes_the_fixture(asan_tools, tmp_path, which, blocks):
    with open(os.path.join(ROOT, "tests", "golden", "sam_small.json")) as f:
        golden = json.load(f)
    write_inputs(golden, tmp_path)
    fl = golden["flags"]
    args = ["-i", "idx", "--genome", "g.fa", "--bucket-len", str(fl["bucket_len"]), "-r", str(fl["read_len"]),
            "-k", str(fl["q"]), "-l", str(fl["k"]), "-s", str(fl["S"]), "-e", str(fl["e"]), "-d", str(fl["d"]),
            "-b", str(fl["b"]), "-n", str(fl["n"]), "-p", str(fl["p"]), "-u", str(fl["u"]), "-f", "1",
            "-q", "reads.fastq", "-o", "out.sam"]
    env = dict(os.environ, ASAN_OPTIONS="protect_shadow_gap=0:detect_leaks=1:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **(blocks or {}))
    r = subprocess.run([asan_tools[which], *args], cwd=str(tmp_path), capture_output=True, text=True, env=env)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and "LeakSanitizer" not in r.stderr, \
        r.stderr[-4000:]
    assert r.returncode == 0, r.stderr[-4000:]
    sq, recs = [], []
    for line in open(tmp_path / "out.sam"):
        f = line.rstrip("\n").split("\t")
        if f[0] == "@SQ":
            sq.append([f[1][3:], int(f[2][3:])])
        elif not line.startswith("@"):
            recs.append([f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5], f[9], f[10]])
    check(golden, which, sq, recs, r.stderr)
