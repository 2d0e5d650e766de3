"""The reference-side binding as a file that compiles: integration/gpu_q_gram_mapper.h is built
against the reference's REAL bucket_map/mapper/mapper.h and used through a `mapper*`.

* not gpu (build container, where /root/reference exists): the translation unit compiles and links against libbmf.so,
  the program starts, and without a device it fails loudly (no CPU fallback);
* gpu: the prebuilt program (it travels like the other built files; the reference does not) maps a FASTQ file and its
  per-bucket lists must be exactly what the C ABI returns for the same windows, scattered in the reference's
  (read, window) order (q_gram_mapper.h:526-533)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF_MAPPER_H = "/root/reference/bucket_map/mapper/mapper.h"
EXE = os.path.join(ROOT, "integration", "_build", "ref_binding")


@pytest.mark.skipif(not os.path.exists(REF_MAPPER_H), reason="the reference tree is only in the build container")
def test_binding_compiles_against_the_reference_mapper_h(tmp_path):
    if os.path.exists(EXE):
        os.remove(EXE)          # compile it NOW (not `make -B`: that would rebuild libbmf.so too)
    subprocess.run(["make", "-C", ROOT, "integration/_build/ref_binding"], check=True, capture_output=True)
    assert os.path.exists(EXE)
    # the TU really saw the reference's header: the class in it is abstract with exactly these members
    src = open(REF_MAPPER_H).read()
    for member in ("virtual void load(", "map(std::filesystem::path const & sequence_file) = 0", "virtual void reset() = 0",
                   "unsigned int num_records"):
        assert member in src
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr
    if not os.path.exists("/dev/kfd"):
        # no device: the constructor must throw (bmf_create -> BMF_ERR_HIP), never fall back to a CPU path
        r = subprocess.run([EXE, "70", "4096", "150", "12", "9", "15", "6", str(tmp_path), "idx", str(tmp_path / "none.fastq")],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "bmf_create" in r.stderr, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("long_reads", [False, True])
def test_binding_lists_equal_the_c_abi(tmp_path, long_reads):
    if not os.path.exists(EXE):
        pytest.skip("integration/_build/ref_binding was not built (it needs /root/reference: build container only)")
    import bucket_map_amd as bma
    from bucket_map_amd import host
    bucket_len, read_len = 4096, 150
    g = host.Genome.synth(31, [300_000, 70_000])
    nb = g.awk_bucket_num(bucket_len)
    ix = host.Index(g, nb, bucket_len, read_len, q=9)
    ix.write(str(tmp_path), "idx")
    rd = host.Reads(g, bucket_len, read_len, 1200 if long_reads else read_len, 300 if long_reads else 2500, sub=0.01, seed=5)
    rd.write_fastq(str(tmp_path / "reads"))
    r = subprocess.run([EXE, str(nb), str(bucket_len), str(read_len), "12", "9", "15", "6", str(tmp_path), "idx",
                        str(tmp_path / "reads.fastq")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Elapsed time for bucket mapping" in r.stderr
    lines = r.stdout.split("\n")
    assert lines[0] == f"num_records {rd.n}"
    assert lines[-2] == "after_reset 0"
    got = {"o": [[] for _ in range(nb)], "r": [[] for _ in range(nb)]}
    for ln in lines[1:-2]:
        t, b, read, pos = ln.split()
        got[t][int(b)].append((int(read), int(pos)))

    flt = bma.Filter(bma.Params.from_cli(nb, read_len=read_len))
    flt.load_index_ptr(ix.rows_ptr, ix.num_rows, ix.k2i_ptr, ix.num_kmers)
    ws, wl, wr, wp = bma.windows_for_reads(rd.offsets, read_len)
    counts, buckets = flt.map_windows(rd.bases, rd.quals, ws, wl)
    flt.close()
    want = {"o": [[] for _ in range(nb)], "r": [[] for _ in range(nb)]}
    for w in range(len(ws)):
        for s, t in enumerate("or"):
            for b in buckets[w, s, :counts[w, s]]:
                want[t][int(b)].append((int(wr[w]), int(wp[w])))
    assert got == want
    assert sum(len(v) for v in want["o"]) + sum(len(v) for v in want["r"]) > 0.9 * rd.n
