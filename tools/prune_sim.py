#!/usr/bin/env python3
"""CPU simulation of the exact-pruning first pass on a synthetic genome: how many 128-bucket chunks
of an item stay alive after a first pass that reads only r rows per sample, from the index itself or from its
2-/4-fold folded copy, with the sample's rows taken far apart ("far", round 2) or sparsest first ("sparse").

    python tools/prune_sim.py --profile genome --total-bp 1701312507 --reads 1200

Design evidence only (numpy over the host indexer's rows and the C oracle's sampled hashes); nothing here is on
the product path.  One JSON line per strategy.
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


def revcomp(h, k):
    rc = 0
    for _ in range(k):
        rc = (rc << 2) | ((~h) & 3)
        h >>= 2
    return rc


def far_order(G):
    order, left = [0], list(range(1, G))
    if G > 1:
        order.append(G - 1)
        left.remove(G - 1)
    while left:
        best = max(left, key=lambda x: min(abs(x - o) for o in order))
        order.append(best)
        left.remove(best)
    return order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", default="genome")
    ap.add_argument("--total-bp", type=int, default=1_701_312_507)
    ap.add_argument("--bucket-len", type=int, default=65536)
    ap.add_argument("--read-len", type=int, default=300)
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--params", default="default", choices=["default", "bench"])
    args = ap.parse_args()
    import bench
    from bucket_map_amd import host
    from oracle import oracle_c

    t0 = time.perf_counter()
    lens = bench.egu_like_record_lengths(args.total_bp)
    g = host.Genome.synth(20240001, lens, 0, profile=args.profile)
    nb = g.awk_bucket_num(args.bucket_len)
    q = 9
    ix = host.Index(g, nb, args.bucket_len, args.read_len, q=q)
    rows = ix.rows()
    W = rows.shape[1]
    print(f"# genome + index: NB={nb} W={W} ({time.perf_counter() - t0:.0f}s)", file=sys.stderr)
    if args.params == "bench":
        k, S, e, b = 14, 20, 0.6, 10
    else:
        k, S, e, b = 12, 15, 0.4, 25
    cli = dict(index_seed=q, query_seed=k, read_len=args.read_len, mapper_samples=S, max_error_rate=e, distinguishability=0.5,
               average_base_quality=b)
    p = oracle_c.params_from_cli(nb, **cli)
    F, G = p.num_fault, k - q + 1
    ora = oracle_c.Index(p, rows_ptr=ix.rows_ptr, n_rows=ix.num_rows, k2i_ptr=ix.k2i_ptr, n_kmers=ix.num_kmers)
    zeros = ora.zeros().astype(np.int64)
    dens = 1.0 - zeros / nb
    reads = host.Reads(g, args.bucket_len, args.read_len, args.read_len, args.reads, seed=20240003)
    n_chunks = (W + 15) // 16
    qmask = 4 ** q - 1
    strategies = [(order, f, r) for order in ("far", "sparse", "sparsefar") for f in (1, 2, 4) for r in range(1, G + 1)]
    live = {s: [] for s in strategies}
    live_m = {s: [] for s in strategies}     # chunks holding a group with LB <= min(m*, F-1): what an adaptive threshold must recount
    live_w = {s: [] for s in strategies}     # chunks with LB <= min(L + 2, F-1), L = the item's lowest lower bound
    fallback = {s: [] for s in strategies}   # ... and whether that window misses a needed chunk
    final_live, d_min, d_all = [], [], []
    far = far_order(G)
    for rd in range(reads.n):
        o0, o1 = int(reads.offsets[rd]), int(reads.offsets[rd + 1])
        _, _, smp, ng = ora.query_sequence(reads.bases[o0:o1][:args.read_len], reads.quals[o0:o1][:args.read_len])
        if ng < 0.2 * S:
            continue
        for strand in (0, 1):
            hs = [int(h) if strand == 0 else revcomp(int(h), k) for h in smp]
            ids = np.array([[(h >> (2 * i)) & qmask for i in range(G)] for h in hs])          # [S, G] row ids (-f 1)
            bits = np.unpackbits(rows[ids.reshape(-1)], axis=1, bitorder="little")[:, :nb].reshape(S, G, nb).astype(bool)
            dd = dens[ids]
            d_all.extend(dd.reshape(-1))
            d_min.extend(dd.min(axis=1))
            exact = (~bits.all(axis=1)).sum(axis=0)                                          # misses per bucket
            alive = exact < F
            m_star = min(int(exact.min()), F - 1)
            final_live.append(len(np.unique(np.nonzero(alive)[0] >> 7)))
            for order in ("far", "sparse", "sparsefar"):
                perm = np.tile(far, (S, 1)) if order == "far" else np.argsort(dd, axis=1, kind="stable")
                if order == "sparsefar":      # sparsest first; then the sparsest among the rows not adjacent to it, then the rest
                    for s_ in range(S):
                        first = perm[s_, 0]
                        rest = [x for x in perm[s_, 1:]]
                        nonadj = [x for x in rest if abs(int(x) - int(first)) >= 2]
                        if nonadj:
                            rest.remove(nonadj[0])
                            rest.insert(0, nonadj[0])
                        perm[s_, 1:] = rest
                ob = np.take_along_axis(bits, perm[:, :, None], axis=1)
                for f in (1, 2, 4):
                    ng_ = (nb + f - 1) // f
                    pad = ng_ * f - nb
                    fb = np.pad(ob, ((0, 0), (0, 0), (0, pad))).reshape(S, G, ng_, f).any(axis=3) if f > 1 else ob
                    acc = np.ones((S, ng_), bool)
                    for r in range(1, G + 1):
                        acc &= fb[:, r - 1]
                        misses = (~acc).sum(axis=0)
                        al = np.nonzero(misses < F)[0] * f                                   # first bucket of each live group
                        live[(order, f, r)].append(len(np.unique(al >> 7)))
                        L = int(misses.min())
                        live_m[(order, f, r)].append(len(np.unique((np.nonzero(misses <= m_star)[0] * f) >> 7)))
                        live_w[(order, f, r)].append(len(np.unique((np.nonzero(misses <= min(L + 2, F - 1))[0] * f) >> 7)))
                        fallback[(order, f, r)].append(m_star > L + 2)
    n = len(final_live)
    print(json.dumps({"items": n, "NB": nb, "F": F, "G": G, "S": S, "density_all_rows_met": float(np.mean(d_all)),
                      "density_sparsest_row_of_a_sample": float(np.mean(d_min)),
                      "chunks_alive_after_exact_count": {"mean": float(np.mean(final_live)), "p50": float(np.median(final_live)),
                                                         "p99": float(np.quantile(final_live, 0.99)),
                                                         "gt16": float((np.array(final_live) > 16).mean())}}))
    sector = 64.0
    for s in strategies:
        lv = np.array(live[s])
        order, f, r = s
        pass1 = S * r * (n_chunks * 16) / f
        # recount: a thin pass (one unseen row per sample) for every live chunk, the full S*G for ~ the truly alive ones
        cost = pass1 + lv.mean() * S * sector + np.mean(final_live) * S * G * sector
        print(json.dumps({"order": order, "fold": f, "rows": r, "pass1_bytes": pass1, "live_mean": float(lv.mean()),
                          "live_p50": float(np.median(lv)), "live_p90": float(np.quantile(lv, 0.9)), "gt16": float((lv > 16).mean()),
                          "gt32": float((lv > 32).mean()), "model_bytes_per_item": float(cost),
                          "adaptive_live_mean": float(np.mean(live_m[s])), "adaptive_gt16": float((np.array(live_m[s]) > 16).mean()),
                          "window_live_mean": float(np.mean(live_w[s])), "window_gt16": float((np.array(live_w[s]) > 16).mean()),
                          "window_p90": float(np.quantile(live_w[s], 0.9)), "window_fallback": float(np.mean(fallback[s]))}))


if __name__ == "__main__":
    main()
