#!/usr/bin/env python3
"""bmf_map_windows (host buffers in and out) at several piece sizes, default and pruned kernels: where the
PCIe-inclusive time goes.  python tools/pcie_probe.py [reads]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))
import numpy as np
import bench
import bucket_map_amd as bma
from bucket_map_amd import host

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
genome = host.Genome.synth(20240001, bench.egu_like_record_lengths(1_701_312_507), 16)
nb = genome.awk_bucket_num(65536)
reads = host.Reads(genome, 65536, 300, 300, n_reads, seed=20240003, threads=16)
flat, _ = genome.flat()
bstart, blen = genome.bucket_views(65536, 300)
ws, wl, _, _ = bma.windows_for_reads(reads.offsets, 300)
pb, pq = bma.pinned_copy(reads.bases), bma.pinned_copy(reads.quals)
out = (np.zeros((len(ws), 2), np.uint32), np.zeros((len(ws), 2, 30), np.uint32))
# raw H2D rate of the link: one pinned copy of the reads
for flags, name in ((0, "default"), (bma.BMF_FLAG_EARLY_EXIT, "pruned")):
    f = bma.Filter(bma.Params.from_cli(nb, read_len=300, flags=flags))
    f.build_index(flat, bstart, blen, host.select_qgrams(9))
    b = f.batch(reads.bases, reads.quals, ws, wl)
    b.run(); f.sync()
    t = time.perf_counter()
    for _ in range(5):
        b.run()
    f.sync()
    dev = (time.perf_counter() - t) / 5
    b.close()
    for piece in (16384, 32768, 65536, 131072, 262144, 1 << 24):
        os.environ["BMF_PIECE_WINDOWS"] = str(piece)
        f.map_windows(pb.array, pq.array, ws, wl, out=out)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            f.map_windows(pb.array, pq.array, ws, wl, out=out)
            best = min(best, time.perf_counter() - t)
        print(f"{name}: device-resident step {dev * 1e3:.2f} ms; host buffers, pieces of {piece}: {best * 1e3:.2f} ms", flush=True)
    f.close()
