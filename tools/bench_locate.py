#!/usr/bin/env python3
"""The locator leg of bench.py on its own (what tools/profile_locate_bench.sh profiles): the filter's candidates for one
batch of the workload, then bml_locate over all of them.  Prints bench.py's `locator` object as one JSON line.

    python tools/bench_locate.py --workload egu --genome-profile genome --calls 3
"""
import argparse
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="egu", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--genome-profile", default="genome", choices=["uniform", "genome"])
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=0.0)
    args = ap.parse_args()
    from bucket_map_amd import host
    total_bp, bucket_len, read_len, n_reads = bench.WORKLOADS[args.workload]
    cli = bench.cli_params("default", read_len)
    inp = bench.Inputs(args.workload, total_bp, bucket_len, read_len, args.reads or n_reads, args.genome_profile, bench.usable_cores())
    k2i = host.select_qgrams(cli["index_seed"], 1.0)
    flt = inp.new_filter(cli, 0, 0, k2i)
    batch = inp.batch(flt)
    batch.run()
    flt.sync()
    counts, buckets = batch.download()
    batch.close()
    flt.close()
    leg = bench.locator_leg(inp, cli, 0, counts, buckets, args.calls, args.cpu_seconds,
                            lambda m: print(f"[bench_locate] {m}", file=sys.stderr, flush=True))
    print(json.dumps(leg), flush=True)


if __name__ == "__main__":
    main()
