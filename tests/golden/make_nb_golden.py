#!/usr/bin/env python3
"""Pins BM_BUCKET_NUM with a reference-owned artefact: runs the reference's OWN script
`/root/reference/get_num_buckets.sh` (bash + awk, the same rule as bucket_map/CMakeLists.txt:13-46) on small FASTA
files and stores what it printed in tests/golden/bucket_num_ref.json.

    python tests/golden/make_nb_golden.py          # build container only: /root/reference does not travel

The FASTA files are not committed: `fasta_text(case)` below rebuilds each one byte for byte from the case's
description (record lengths, line width, decorations), so the test (tests/test_host.py::
test_bucket_num_matches_the_reference_script) feeds the SAME bytes to this repository's reader and
`awk_bucket_num` and compares with the committed numbers.  This pins NB only -- the oracles stay "parity unpinned".
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SCRIPT = "/root/reference/get_num_buckets.sh"

# (name, bucket_len, line width, [(record length, decoration)]).  Decorations: "" plain; "lower" lower case;
# "n" every 7th base an N; "blank" a blank line inside the record; "desc" a header with a description.
CASES = [
    ("exact_multiple", 1024, 60, [(2048, ""), (1024, "")]),                 # len / bucket_len is an integer: no +1
    ("dropped_tail", 1024, 60, [(2048 + 50, ""), (1024 + 100, "")]),        # tails of 50 and 100 <= read_len: counted here, cut
                                                                            # by utils.h:88-90 -> NB exceeds the kept buckets
    ("one_base_over", 4096, 70, [(4097, ""), (4095, ""), (1, "")]),
    ("empty_record", 1024, 60, [(0, ""), (3000, ""), (0, "")]),             # `if (seqlen)`: header-only records add nothing
    ("decorated", 512, 61, [(5000, "lower"), (777, "n"), (1300, "blank"), (2049, "desc")]),
    ("many_small", 65536, 60, [(n, "") for n in (10_000, 210_000, 65_536, 65_537, 131_072, 300, 301)]),
    ("single_line_records", 100, 100_000, [(12_345, ""), (100, ""), (99, ""), (101, "")]),
]


def fasta_text(case):
    name, _, width, records = case
    out = []
    for i, (n, deco) in enumerate(records):
        out.append(f">{name}_{i}" + (" some description len=%d" % n if deco == "desc" else ""))
        seq = "".join("ACGT"[(j * 7 + i + (j >> 3)) & 3] for j in range(n))
        if deco == "lower":
            seq = seq.lower()
        elif deco == "n":
            seq = "".join("N" if j % 7 == 3 else c for j, c in enumerate(seq))
        lines = [seq[j:j + width] for j in range(0, n, width)]
        if deco == "blank" and len(lines) > 2:
            lines.insert(2, "")
        out.extend(lines)
    return "\n".join(out) + "\n"


def main():
    if not os.path.exists(REF_SCRIPT):
        sys.exit(f"{REF_SCRIPT} is not here: run this in the build container")
    golden = {"source": "stdout of /root/reference/get_num_buckets.sh <fasta> <bucket_len> (bash + awk), run by "
                        "tests/golden/make_nb_golden.py", "cases": {}}
    with tempfile.TemporaryDirectory() as d:
        for case in CASES:
            path = os.path.join(d, case[0] + ".fa")
            with open(path, "w") as f:
                f.write(fasta_text(case))
            r = subprocess.run(["bash", REF_SCRIPT, path, str(case[1])], capture_output=True, text=True, check=True)
            golden["cases"][case[0]] = {"bucket_len": case[1], "bucket_num": int(r.stdout.strip())}
    with open(os.path.join(HERE, "bucket_num_ref.json"), "w") as f:
        json.dump(golden, f, indent=1)
        f.write("\n")
    print(json.dumps(golden["cases"]))


if __name__ == "__main__":
    main()
