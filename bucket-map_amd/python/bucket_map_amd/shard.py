"""Read sharding across GPUs / ranks and the order-preserving merge of per-shard results.

The filter shards with no exchange step: windows are independent, the index is
replicated.  Shards are CONTIGUOUS window ranges, so concatenating per-shard outputs in rank order
restores the global (read, window) order that q_gram_mapper::map's per-bucket lists rely on
(bucket_map/mapper/q_gram_mapper.h:526-533).  The C++ host wrapper (gpu_q_gram_mapper.h) applies the
same rule across the devices of one process; this module is the multi-process form used by bench.py
and covered by tests/test_sharding.py on gloo.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Windows [lo, hi) owned by `rank`: the same floor(n*r/world) split as the C++ wrapper."""
    return n * rank // world, n * (rank + 1) // world


def merge_shards(parts):
    """parts: per-rank (counts[n_r,2], buckets[n_r,2,mc]) in rank order -> global arrays."""
    counts = np.concatenate([p[0] for p in parts], axis=0)
    buckets = np.concatenate([p[1] for p in parts], axis=0)
    return counts, buckets


def segments_from_results(counts, buckets, win_read, win_pos, num_buckets: int):
    """The scatter of q_gram_mapper::map (q_gram_mapper.h:526-533): per bucket, the (read, window start)
    pairs in (read, window) order, for the read as-is and for its reverse complement."""
    orig = [[] for _ in range(num_buckets)]
    rev = [[] for _ in range(num_buckets)]
    for w in range(len(counts)):
        seg = (int(win_read[w]), int(win_pos[w]))
        for b in buckets[w, 0, : counts[w, 0]]:
            orig[int(b)].append(seg)
        for b in buckets[w, 1, : counts[w, 1]]:
            rev[int(b)].append(seg)
    return orig, rev
