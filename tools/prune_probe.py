#!/usr/bin/env python3
"""The exact-pruning kernels (BMF_FLAG_EARLY_EXIT) under forced configurations, on one workload (GPU box).

    python tools/prune_probe.py --genome-profile genome --configs "auto;BMF_PASS1_ROWS=0;BMF_FOLD=2,BMF_FOLD_ROWS=2"

Builds the genome, the index (on the device) and the reads once; then, for every configuration (a comma-separated
list of the experiment variables of DESIGN.md section 6, `auto` = the library's own choice, `plain` = the flag off), times
`--steps` runs of the device-resident batch, reports the step time, the items that went through the recount and the
slow kernels, and checks the outputs against the plain kernel's.  One line of JSON per configuration.
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
    ap.add_argument("--workload", default="egu")
    ap.add_argument("--genome-profile", default="genome")
    ap.add_argument("--params", default="default", choices=["default", "bench"])
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--configs", default="plain;auto")
    args = ap.parse_args()
    import bench
    import bucket_map_amd as bma
    from bucket_map_amd import host

    total_bp, bucket_len, read_len, _ = bench.WORKLOADS[args.workload]
    cli = (dict(index_seed=9, query_seed=14, read_len=read_len, mapper_samples=20, max_error_rate=0.6, distinguishability=0.5,
                average_base_quality=10) if args.params == "bench" else
           dict(index_seed=9, query_seed=12, read_len=read_len, mapper_samples=15, max_error_rate=0.4, distinguishability=0.5,
                average_base_quality=25))
    threads = bench.usable_cores()
    genome = host.Genome.synth(20240001, bench.workload_record_lengths(args.workload, total_bp), threads, profile=args.genome_profile)
    nb = genome.awk_bucket_num(bucket_len)
    reads = host.Reads(genome, bucket_len, read_len, read_len, args.reads, sub=0.002, ins=0.00025, dele=0.00025, seed=20240003,
                       threads=threads)
    flat, _ = genome.flat()
    bstart, blen = genome.bucket_views(bucket_len, read_len)
    k2i = host.select_qgrams(9)
    ws, wl, _, _ = bma.windows_for_reads(reads.offsets, read_len)
    want = None
    for cfg in args.configs.split(";"):
        cfg = cfg.strip()
        env = {} if cfg in ("auto", "plain") else dict(kv.split("=") for kv in cfg.split(","))
        os.environ.update(env)
        try:
            flt = bma.Filter(bma.Params.from_cli(nb, flags=0 if cfg == "plain" else bma.BMF_FLAG_EARLY_EXIT, **cli))
            flt.build_index(flat, bstart, blen, k2i)
            batch = flt.batch(reads.bases, reads.quals, ws, wl)
            batch.run()
            batch.run()          # (the guard / the tuner act on the first runs)
            flt.sync()
            flt.profile_begin(args.steps)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                batch.run()
            flt.sync()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            ms_sample, ms_rest = flt.profile_end(args.steps)
            c, b = batch.download()
            if want is None:
                want = (c, b)
            mask = np.arange(b.shape[-1])[None, None, :] < want[0][:, :, None]
            same = bool(np.array_equal(c, want[0]) and np.array_equal(b[mask], want[1][mask]))
            info = flt.info()
            out = {"config": cfg, "ms_per_step": round(ms, 3), "ms_sample": round(float(np.mean(ms_sample)), 3),
                   "ms_after_sample": round(float(np.mean(ms_rest)), 3), "identical": same,
                   "pass1_rows": info["pass1_rows"], "fold": info["pass1_fold"], "fold_rows": info["pass1_fold_rows"]}
            if info["pass1_rows"]:
                out["items_recounted"], out["items_slow"] = batch.pass2_counts()
                out["recount_loads"] = batch.recount_loads()
            print(json.dumps(out), flush=True)
            batch.close()
            flt.close()
        finally:
            for k in env:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
