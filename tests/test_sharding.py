"""The N>1 path on CPU: two gloo ranks shard the windows, each maps its shard (the CPU oracle stands in
for the device), rank 0 gathers and merges -- the result must equal the single-process result, and the
per-bucket (read, window) lists must come out in the reference's order.  No collective is on the data
path in production either; torch.distributed only moves the results here."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import bucket_map_amd as bma
    from bucket_map_amd import shard
    from conftest import Case

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank builds the same (replicated) index and the same reads, deterministically
    case = Case(record_lengths=[120_000], bucket_len=2048, read_len=150, n_reads=90, sim_read_len=700, seed=31)
    rd = case.reads
    ws, wl, rid, pos = bma.windows_for_reads(rd.offsets, case.read_len)     # 5 windows per long read
    lo, hi = shard.shard_range(len(ws), rank, world)
    ix = case.oracle_index()
    c, b, _ = ix.map_windows(rd.bases, rd.quals, ws[lo:hi], wl[lo:hi])
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((c, b), gathered, dst=0)
    if rank == 0:
        counts, buckets = shard.merge_shards(gathered)
        c1, b1, _ = ix.map_windows(rd.bases, rd.quals, ws, wl)               # single-process truth
        ok = np.array_equal(counts, c1)
        mask = np.arange(b1.shape[-1])[None, None, :] < c1[:, :, None]
        ok = ok and np.array_equal(buckets[mask], b1[mask])
        orig, rev = shard.segments_from_results(counts, buckets, rid, pos, case.num_buckets)
        ordered = all(lst == sorted(lst) for lst in orig + rev)             # (read, window) order per bucket
        nonempty = sum(len(x) for x in orig + rev)
        with open(out_path, "w") as f:
            f.write(f"{int(ok)} {int(ordered)} {nonempty} {len(ws)}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    import torch.multiprocessing as mp     # imported here: keeps torch out of the GPU test process
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ok, ordered, nonempty, n = (int(x) for x in open(out).read().split())
    assert ok == 1, "merged shards differ from the single-process result"
    assert ordered == 1, "per-bucket lists lost their (read, window) order"
    assert n == 5 * 90 and nonempty > 200


def _locator_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from bucket_map_amd import shard
    from oracle import oracle_c as oc
    from test_locator import make_case

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank holds the genome and all sampled k-mers (replicated) and scans ITS contiguous range of the candidates,
    # which arrive grouped by bucket: a bucket whose run straddles the cut is scanned by both ranks
    case = make_case(np.random.default_rng(5), n_buckets=7, bucket_len=2048, read_len=150, n_reads=150)
    n = len(case["pb"])
    lo, hi = shard.shard_range(n, rank, world)
    off, votes = oc.locate(case["k"], case["p"], 4, 6, case["genome"], case["bstart"], case["blen"], case["sh"], case["sp"],
                           case["sl"], case["pb"][lo:hi], case["pw"][lo:hi], case["pr"][lo:hi])
    # the verifier's split: alignments of the located candidates, cut by summed cell count
    keep = np.nonzero(off > 0)[0]
    ts = (case["bstart"][case["pb"][lo:hi][keep]] + off[keep].astype(np.uint64)).astype(np.uint64)
    tl = np.full(len(keep), 150 + 1 + 3, np.uint32)
    reads = np.concatenate([case["genome"][int(s):int(s) + 150] for s in ts]) if len(keep) else np.zeros(0, np.uint8)
    qs = (np.arange(len(keep)) * 150).astype(np.uint64)
    ql = np.full(len(keep), 150, np.uint32)
    sc, bg, co, cg = oc.align_batch(case["genome"], reads, ts, tl, np.zeros(len(keep), np.uint8), qs, ql)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((off, votes, sc, bg), gathered, dst=0)
    if rank == 0:
        off_all = np.concatenate([g[0] for g in gathered])
        votes_all = np.concatenate([g[1] for g in gathered])
        o1, v1 = oc.locate(case["k"], case["p"], 4, 6, case["genome"], case["bstart"], case["blen"], case["sh"], case["sp"],
                           case["sl"], case["pb"], case["pw"], case["pr"])
        ok = np.array_equal(off_all, o1) and np.array_equal(votes_all, v1)
        straddles = int(case["pb"][shard.shard_range(n, 1, world)[0] - 1] == case["pb"][shard.shard_range(n, 1, world)[0]])
        perfect = all((g[2] == 0).all() and (g[3] == 0).all() for g in gathered)   # a read cut from the text aligns at 0, score 0
        with open(out_path, "w") as f:
            f.write(f"{int(ok)} {straddles} {int((o1 > 0).sum())} {int(perfect)}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_locator_scan_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_locator_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ok, straddles, located, perfect = (int(x) for x in open(out).read().split())
    assert ok == 1, "candidate ranges scanned by two ranks differ from the single-process scan"
    assert straddles == 1, "the cut was meant to fall inside one bucket's run of candidates"
    assert located > 100 and perfect == 1


def test_shard_ranges_partition():
    sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))
    from bucket_map_amd import shard
    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
