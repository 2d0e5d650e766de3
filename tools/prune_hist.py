#!/usr/bin/env python3
"""Histogram of what pass 1 of the two-pass pruning leaves per item (stored chunks, chunks at the lowest level) on the Egu
geometry, uniform and genome-like genome.  python tools/prune_hist.py [uniform|genome] [reads]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bucket_map_amd as bma  # noqa: E402
from bucket_map_amd import host  # noqa: E402

profile = sys.argv[1] if len(sys.argv) > 1 else "genome"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
total_bp, bucket_len, read_len, _ = bench.WORKLOADS["egu"]
cli = bench.cli_params("default", read_len)
inp = bench.Inputs("egu", total_bp, bucket_len, read_len, n_reads, profile, bench.usable_cores())
flt = inp.new_filter(cli, 0, bma.BMF_FLAG_EARLY_EXIT, host.select_qgrams(9, 1.0))
b = inp.batch(flt)
b.run(); flt.sync(); b.run(); flt.sync()
print(profile, flt.info(), "recounted/slow", b.pass2_counts(), "loads", b.recount_loads())
st, lo = b.live_histogram()
tot = float(st.sum())
print("stored chunks n: share of items")
print(" ".join(f"{i}:{100 * st[i] / tot:.1f}" for i in range(34) if st[i]))
print("chunks at the lowest level: share of items")
print(" ".join(f"{i}:{100 * lo[i] / tot:.1f}" for i in range(34) if lo[i]))
