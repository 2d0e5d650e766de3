#!/usr/bin/env python3
"""Throughput of the alignment verifier (include/bmv.h) on a synthetic batch shaped like `bucketmap_align`'s
work on BASELINE configs[1]: one located candidate per read, text window = read + 1 + 2 % (bucket_locator.h:550).

    python tools/bench_verify.py [--reads 1000000] [--len 300] [--indel-rate 0.02] [--cpu-sample 2000]

Prints one JSON line: alignments/s and cell updates/s of the device kernels (HIP events inside bmv_align),
the wall time of the call (host buffers in, results out), and the CPU restatement (oracle, full DP matrix,
1 core) on a sample beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=300)
    ap.add_argument("--indel-rate", type=float, default=0.02)
    ap.add_argument("--sub", type=float, default=0.002)
    ap.add_argument("--cpu-sample", type=int, default=2000)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--mixed", type=int, default=0,
                    help="query lengths log-uniform in [MIXED, --len] (forward strand only): a batch of several length classes")
    args = ap.parse_args()

    import torch  # noqa: F401  (HIP runtime first, as in bench.py)
    from bucket_map_amd import verify

    rng = np.random.default_rng(20240003)
    m = args.len
    width = m + 1 + int(np.float32(args.indel_rate) * np.float32(m))
    genome = rng.integers(0, 4, 64 << 20, dtype=np.uint8)
    genome = np.frombuffer(b"ACGT", np.uint8)[genome]
    start = rng.integers(0, len(genome) - width - 8, args.reads).astype(np.uint64)
    rc = rng.integers(0, 2, args.reads).astype(np.uint8) * (0 if args.mixed else 1)
    # reads: the window's bases from offset 1 (so begin = 1), substitutions only + strand flips, built vectorised
    idx = start[:, None] + 1 + np.arange(m, dtype=np.uint64)[None, :]
    reads = genome[idx]
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    flip = rc.astype(bool)
    reads[flip] = comp[reads[flip]][:, ::-1]
    subs = rng.random(reads.shape) < args.sub
    reads[subs] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(subs.sum()))]
    reads = np.ascontiguousarray(reads).reshape(-1)
    qs = (np.arange(args.reads, dtype=np.uint64) * m)
    ql = np.full(args.reads, m, np.uint32)
    tl = np.full(args.reads, width, np.uint32)
    if args.mixed:                                     # each read keeps its first ql bases
        ql = np.exp(rng.uniform(np.log(args.mixed), np.log(m), args.reads)).astype(np.uint32)
        tl = (ql + 1 + (np.float32(args.indel_rate) * ql.astype(np.float32)).astype(np.uint32)).astype(np.uint32)

    v = verify.Verifier()
    v.load_genome(genome)
    best_ms, best_wall = None, None
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        score, begin, off, cg = v.align(reads, start, tl, rc, qs, ql)
        wall = time.perf_counter() - t0
        st = v.stats()
        if best_ms is None or st["ms_kernels"] < best_ms:
            best_ms, best_wall = st["ms_kernels"], wall
    cells = st["cells"]

    ns = min(args.cpu_sample, args.reads)
    same, cpu = None, None
    if ns > 0:                                         # (--cpu-sample 0: no oracle in the process at all -- bench.py's leg)
        from oracle import oracle_c as oc
        t0 = time.perf_counter()
        s_ref, b_ref, o_ref, c_ref = oc.align_batch(genome, reads, start[:ns], tl[:ns], rc[:ns], qs[:ns], ql[:ns])
        cpu_s = time.perf_counter() - t0
        same = bool(np.array_equal(score[:ns], s_ref) and np.array_equal(begin[:ns], b_ref) and
                    np.array_equal(off[: ns + 1], o_ref) and np.array_equal(cg[: int(o_ref[ns])], c_ref))
        cpu = {"value": ns / cpu_s, "unit": "alignments/s", "cores": 1, "kind": "port", "sample": f"first {ns} alignments"}
    print(json.dumps({
        "metric": "verified alignments/s (device kernels)", "value": args.reads / (best_ms * 1e-3), "unit": "alignments/s",
        "config": {"alignments": args.reads, "query_len": m, "text_len": width, "mixed_from": args.mixed},
        "ms_kernels": best_ms, "cell_updates_per_s": cells / (best_ms * 1e-3), "wall_s_host_buffers": best_wall,
        "mean_edits": float(-score.mean()), "cigar_entries": int(len(cg)),
        "cpu_baseline": cpu,
        "checks": {"sample_identical_to_oracle": same},
    }))


if __name__ == "__main__":
    main()
