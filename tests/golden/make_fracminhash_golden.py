#!/usr/bin/env python3
"""Pins the FracMinHash q-gram selection with the reference's OWN header.

    python tests/golden/make_fracminhash_golden.py     # build container only: /root/reference does not travel

`/root/reference/bucket_map/tools/hash_function_generator.h` includes only <utility> <functional> <ctime> <cstdlib>
and compiles as it lies.  The driver below (the only C++ written here) constructs the generator, re-seeds the C
library's rand() with a FIXED seed (the constructor seeds it with time()), calls `generate(10000)` as
`bucket_map/main.cpp:176-178` does and then applies the indexer's selection rule -- `min_hash_function(i) <=
threshold`, rows numbered in ascending q-gram (`bucket_indexer.h:147-157`), threshold `(unsigned)(10000 * f)` in
float32 (`main.cpp:185`) -- with the reference's closure itself.  The closure hides x, y, p, so the driver re-seeds
once more and draws the two rand() values `generate` drew (glibc's rand() is a pure function of the seed), and asks
the reference's own `choose_prime_larger_than` for p.  Stored per (seed, q, f): x, y, p, the number of kept
q-grams, the first 24 of them, 16 raw hash values and the SHA-256 of the whole int32 kmer_to_index array.

tests/test_host.py::test_fracminhash_matches_the_reference_header feeds (x, y, p) to this repository's
`select_qgrams` (bucket-map_amd/host/bm_indexer.h) and compares.  This pins row f3's selection only -- the hot path
stays "parity unpinned".
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_HEADER = "/root/reference/bucket_map/tools/hash_function_generator.h"

DRIVER = r"""
#include "%s"
#include <cstdio>
#include <cstdint>
#include <vector>
int main(int argc, char **argv) {
    unsigned seed = (unsigned)std::strtoul(argv[1], nullptr, 10);
    unsigned q = (unsigned)std::strtoul(argv[2], nullptr, 10);
    float frac = std::strtof(argv[3], nullptr);
    std::size_t HASH_TABLE_SIZE = 10000;
    hash_function_generator gen;
    std::srand(seed);
    auto min_hash_function = gen.generate(HASH_TABLE_SIZE);
    unsigned min_hash_threshold = (unsigned int)(HASH_TABLE_SIZE * frac);
    std::size_t p = gen.choose_prime_larger_than(10 * HASH_TABLE_SIZE);
    std::srand(seed);
    std::size_t x = std::rand() %% (p - 1) + 1, y = std::rand() %% p;
    std::printf("%%zu %%zu %%zu %%u\n", x, y, p, min_hash_threshold);
    std::vector<int32_t> kmer_to_index;
    int index = 0;
    for (unsigned int i = 0; i < (1u << (2 * q)); i++) {
        if (min_hash_function(i) <= min_hash_threshold) { kmer_to_index.push_back(index); index++; }
        else kmer_to_index.push_back(-1);
    }
    for (unsigned i = 0; i < 16; i++) std::printf("%%zu ", min_hash_function(i * 9973u + 5u));
    std::printf("\n");
    std::fwrite(kmer_to_index.data(), sizeof(int32_t), kmer_to_index.size(), stderr);
    return 0;
}
"""

# (seed, q, kmer_frac): the CLI's -f values the reference's scripts use (0.25 short-read sweep, 0.5, 1) and an odd one
CASES = [(1, 9, 0.25), (1, 9, 0.5), (1, 9, 1.0), (20240004, 9, 0.25), (20240004, 9, 0.5), (7, 8, 0.1),
         (123456789, 10, 0.25), (42, 6, 0.75)]


def main():
    if not os.path.exists(REF_HEADER):
        sys.exit(f"{REF_HEADER} is not here: run this in the build container")
    golden = {"source": "the closure returned by hash_function_generator::generate(10000) of " + REF_HEADER +
                        " after std::srand(seed), run by tests/golden/make_fracminhash_golden.py (glibc rand())",
              "sample_inputs": [i * 9973 + 5 for i in range(16)], "cases": []}
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "drv.cpp")
        with open(src, "w") as f:
            f.write(DRIVER % REF_HEADER)
        exe = os.path.join(d, "drv")
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], check=True)
        for seed, q, frac in CASES:
            r = subprocess.run([exe, str(seed), str(q), repr(frac)], capture_output=True, check=True)
            lines = r.stdout.decode().splitlines()
            x, y, p, thr = (int(v) for v in lines[0].split())
            import numpy as np
            k2i = np.frombuffer(r.stderr, dtype=np.int32)
            assert k2i.size == 4 ** q
            kept = np.flatnonzero(k2i >= 0)
            golden["cases"].append({"seed": seed, "q": q, "kmer_frac": frac, "x": x, "y": y, "p": p, "threshold": thr,
                                    "kept": int(kept.size), "first_kept": kept[:24].tolist(),
                                    "sample_hashes": [int(v) for v in lines[1].split()],
                                    "k2i_sha256": hashlib.sha256(k2i.tobytes()).hexdigest()})
    with open(os.path.join(HERE, "fracminhash_ref.json"), "w") as f:
        json.dump(golden, f, indent=1)
        f.write("\n")
    for c in golden["cases"]:
        print(c["seed"], c["q"], c["kmer_frac"], c["x"], c["y"], c["kept"], c["k2i_sha256"][:16])


if __name__ == "__main__":
    main()
