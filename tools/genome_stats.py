#!/usr/bin/env python3
"""What the candidate-bucket filter sees of a synthetic genome (CPU only: host indexer + C oracle).

    python tools/genome_stats.py --profile genome --total-bp 400000000 --bucket-len 65536 --q 9 --reads 20000

Prints one JSON line: the share of index rows that pass the distinguishability threshold (the reference logs
95.8 % for GRCh38 at bucket_len 65 536, bucket_map/benchmark/short_read/log/bucketmap_3_map.log:8), quantiles of the
row density, the share of simulated reads with a candidate (reference: 94.9 %, :12), candidates per read and
strand (reference: 0.81, :13-14) and how many reads recover their source bucket.  Used to tune bm_synth.h's
genome-like profile; the numbers it printed are quoted in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", default="genome", choices=["genome", "uniform"])
    ap.add_argument("--total-bp", type=int, default=400_000_000)
    ap.add_argument("--records", type=int, default=4)
    ap.add_argument("--bucket-len", type=int, default=65536)
    ap.add_argument("--read-len", type=int, default=300)
    ap.add_argument("--q", type=int, default=9)
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--sigma", type=float, default=0.0)
    ap.add_argument("--p-repeat", type=float, default=-1.0)
    ap.add_argument("--seed", type=int, default=20240001)
    ap.add_argument("--params", default="default", choices=["default", "bench", "long"])
    ap.add_argument("--sub", type=float, default=0.002)
    ap.add_argument("--indel", type=float, default=0.00025)
    ap.add_argument("--sim-read-len", type=int, default=0)
    args = ap.parse_args()
    from bucket_map_amd import host
    import bucket_map_amd as bma
    from oracle import oracle_c

    t0 = time.perf_counter()
    lens = [args.total_bp // args.records] * args.records
    g = host.Genome.synth(args.seed, lens, 0, profile=args.profile, sigma=args.sigma, p_repeat=args.p_repeat)
    t_gen = time.perf_counter() - t0
    nb = g.awk_bucket_num(args.bucket_len)
    t0 = time.perf_counter()
    ix = host.Index(g, nb, args.bucket_len, args.read_len, q=args.q)
    t_ix = time.perf_counter() - t0
    rows = ix.rows()
    # popcount per row
    pop = np.zeros(rows.shape[0], np.int64)
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint16)
    step = 8192
    for r0 in range(0, rows.shape[0], step):
        pop[r0:r0 + step] = lut[rows[r0:r0 + step]].sum(axis=1)
    zeros = nb - pop
    zeros[pop == 0] = nb
    thr = int(np.float32(0.5) * np.float32(nb))
    dens = pop / nb
    out = {"profile": args.profile, "bp": g.total_length(), "gap_bases": g.gap_bases(), "bucket_len": args.bucket_len, "NB": nb,
           "q": args.q, "gen_s": round(t_gen, 2), "index_s": round(t_ix, 2),
           "rows_passing_threshold": float((zeros >= thr).mean()), "rows_empty": float((pop == 0).mean()),
           "density_quantiles": {str(p): float(np.quantile(dens, p)) for p in (0.01, 0.1, 0.5, 0.9, 0.99, 0.999)},
           "density_mean": float(dens.mean()),
           # what a read meets: density weighted by how often a q-gram occurs (sum d^2 / sum d)
           "density_met_by_reads": float((dens * dens).sum() / max(dens.sum(), 1e-30))}
    if args.reads:
        cli = {"default": dict(index_seed=args.q, query_seed=12, read_len=args.read_len, mapper_samples=15, max_error_rate=0.4,
                               distinguishability=0.5, average_base_quality=25),
               "bench": dict(index_seed=args.q, query_seed=14, read_len=args.read_len, mapper_samples=20, max_error_rate=0.6,
                             distinguishability=0.5, average_base_quality=10),
               "long": dict(index_seed=args.q, query_seed=12, read_len=args.read_len, mapper_samples=30, max_error_rate=0.9,
                            distinguishability=0.5, average_base_quality=25)}[args.params]
        srl = args.sim_read_len or args.read_len
        reads = host.Reads(g, args.bucket_len, args.read_len, srl, args.reads, sub=args.sub, ins=args.indel, dele=args.indel)
        ora = oracle_c.Index(oracle_c.params_from_cli(nb, **cli), rows_ptr=ix.rows_ptr, n_rows=ix.num_rows, k2i_ptr=ix.k2i_ptr,
                             n_kmers=ix.num_kmers)
        ws, wl, wr, _ = bma.windows_for_reads(reads.offsets, args.read_len)
        t0 = time.perf_counter()
        c, b, _ = ora.map_windows(reads.bases, reads.quals, ws, wl)
        t_map = time.perf_counter() - t0
        wr = np.asarray(wr)
        n = reads.n
        per_read = np.zeros((n, 2), np.int64)
        np.add.at(per_read, wr, c)
        strand = reads.truth_rc.astype(int)
        hit = np.zeros(n, bool)
        for w in range(len(ws)):
            r = wr[w]
            s = strand[r]
            if reads.truth_bucket[r] in b[w, s, :c[w, s]]:
                hit[r] = True
        out.update({"reads": n, "windows": len(ws), "map_s": round(t_map, 2),
                    "reads_with_candidates": float((per_read.sum(axis=1) > 0).mean()),
                    "candidates_per_read_per_strand": float(per_read.mean()),
                    "source_bucket_recovered": float(hit.mean()),
                    "windows_with_more_than_8_candidates": float((c > 8).mean())})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
