#!/usr/bin/env python3
"""Benchmark of the candidate-bucket filter on MI355X: mapped reads/sec + % of the HBM roofline.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (sample kernel + vote kernel) over one batch of synthetic reads
already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: Egu-like 1.70 Gbp synthetic
genome cut into 65 536-bp buckets, `-f 1` index, 1 M x 300 bp simulated reads (sub 0.002,
ins = del 0.00025), CLI-default parameters (k=12 q=9 S=15 e=0.4 -> F=6).  With N GPUs the index is
replicated and there is no collective on the data path; torch.distributed only provides the barrier and
the max-over-ranks of the time.  N > 1 defaults to BASELINE.json configs[2] -- the SAME 1 M reads cut into
N contiguous shards, one per rank (`--scaling strong`, "scaling": "strong"); `--scaling weak` gives every
rank its own 1 M reads instead, and the strong run reports that leg too, as `weak_scaling`.

Rank 0 prints ONE JSON line.  `roofline` prices the vote kernel: algorithmic bytes = (index rows the
reference ANDs, both orientations) x ceil(NB/8)   / mean kernel time from HIP events on
the kernel's own stream.  `cpu_baseline` = the CPU oracle (a port: the reference cannot be built
here) timed on one host core over a bounded sample of the same reads.

Beside the headline, the default N = 1 run reports more legs on the same card (never as `value`; each is guarded: a leg that
fails is recorded under its key and costs nothing else):
  `locator`               north_star's second replaced subsystem, `bucket_map/locator`'s candidate scan: bml_locate over EVERY
                          candidate the filter produced for the batch -- scan and vote-replay kernel times (HIP events inside the
                          library), occurrences, algorithmic bytes and the fraction of the HBM peak they amount to, the scan's VALU
                          wave-instructions against the issue peak (live rocprofv3 child), the CPU oracle on the candidates of the
                          first buckets beside it (1 thread) and GPU == oracle on that sample.  The `skewed` leg carries its own;
  `skewed`                the same geometry on bm_synth.h's genome-LIKE genome (skewed q-gram spectrum, repeat families,
                          satellites, segmental duplications): what the data-dependent parts of the path do on real data --
                          rows that fail the distinguishability threshold, reads without a candidate, and the
                          exact-pruning kernels, whose form the library measures on the first batch;
  `roofline_large_index`  the vote kernel on a 4.6 GB index (2.29 Gbp at bucket_len 16 384, NB = 140 471), of which the
                          256 MiB Infinity Cache can hold 6 %: the un-flattered HBM fraction (the 872 MB headline index
                          is 31 % cache-resident);
  `verifier`              the alignment verifier (include/bmv.h) on two short-read shapes and the long-read shape, each with a
                          live SQ_INSTS_VALU roofline against the VALU issue peak.
"""
import argparse
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20

WORKLOADS = {
    # name: (total genome bp, bucket_len, read_len, reads per GPU)
    "egu": (1_701_312_507, 65536, 300, 1_000_000),        # BASELINE.json configs[1] (headline)
    "ecoli": (4_641_652, 65536, 150, 10_000),             # configs[0] geometry (plumbing)
    "mini": (40_000_000, 65536, 300, 100_000),            # quick rehearsal
    "grch38": (3_100_000_000, 65536, 150, 1_000_000),     # configs[3] geometry (secondary data point)
}
LARGE_INDEX = dict(total_bp=2_293_760_000, bucket_len=16384)   # NB = 140 471: a 4.63 GB index, three 65 536-bucket slices

# GRCh38 chromosome lengths in Mbp (1..22, X, Y): only their RATIOS are used
GRCH38_MBP = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28,
              114.36, 107.04, 101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82, 156.04, 57.23]


def usable_cores():
    """CPU cores this process may really use: the cgroup quota if there is one, else its affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def egu_like_record_lengths(total):
    """16 records of 100 Mbp + 916 records with lengths uniform in [10 kbp, 210 kbp]
    (seed 20240002) rescaled so that the total is `total`."""
    import numpy as np
    big = [100_000_000] * 16
    if total <= sum(big):
        return [total]
    rng = np.random.default_rng(20240002)
    small = rng.integers(10_000, 210_001, 916).astype(np.float64)
    rest = total - sum(big)
    small = np.floor(small * (rest / small.sum())).astype(np.int64)
    small[-1] += rest - small.sum()
    return big + [int(x) for x in small]


def workload_record_lengths(workload, total_bp):
    if workload == "ecoli":
        return [total_bp]
    if workload == "grch38":
        return [int(total_bp * m / sum(GRCH38_MBP)) for m in GRCH38_MBP]
    return egu_like_record_lengths(total_bp)


def cli_params(params, read_len):
    if params == "bench":       # benchmark/short_read/benchmark_map.sh:31
        return dict(index_seed=9, query_seed=14, read_len=read_len, mapper_samples=20, max_error_rate=0.6, distinguishability=0.5,
                    average_base_quality=10)
    return dict(index_seed=9, query_seed=12, read_len=read_len, mapper_samples=15, max_error_rate=0.4, distinguishability=0.5,
                average_base_quality=25)


class Inputs:
    """Synthetic genome + simulated reads of one workload, identical on every rank except the reads' seed."""

    def __init__(self, workload, total_bp, bucket_len, read_len, n_reads, profile, threads, read_seed=20240003):
        import bucket_map_amd as bma
        from bucket_map_amd import host
        self.workload, self.bucket_len, self.read_len, self.profile = workload, bucket_len, read_len, profile
        t0 = time.perf_counter()
        self.lens = workload_record_lengths(workload, total_bp)
        self.genome = host.Genome.synth(20240001, self.lens, threads, profile=profile)
        self.nb = self.genome.awk_bucket_num(bucket_len)
        self.genome_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.reads = host.Reads(self.genome, bucket_len, read_len, read_len, n_reads, sub=0.002, ins=0.00025, dele=0.00025,
                                seed=read_seed, threads=threads)
        self.reads_s = time.perf_counter() - t0
        self.win_start, self.win_len, _, _ = bma.windows_for_reads(self.reads.offsets, read_len)   # mapper::map's windowing
        self.row_bytes = (self.nb + 7) >> 3

    def new_filter(self, cli, device, flags, k2i):
        """A filter context with the `-f` index built on the device (bmf_build_index: byte-identical to the host indexer,
        tests/test_index_build_gpu.py)."""
        import bucket_map_amd as bma
        flt = bma.Filter(bma.Params.from_cli(self.nb, device=device, flags=flags, **cli))
        flat, _ = self.genome.flat()
        bstart, blen = self.genome.bucket_views(self.bucket_len, self.read_len)
        flt.build_index(flat, bstart, blen, k2i)
        return flt

    def batch(self, flt, shard=None):
        import numpy as np
        rd = self.reads
        if shard is None or (shard.start == 0 and shard.stop == rd.n):
            return flt.batch(rd.bases, rd.quals, self.win_start, self.win_len)
        lo, hi = int(rd.offsets[shard.start]), int(rd.offsets[shard.stop])
        return flt.batch(rd.bases[lo:hi], rd.quals[lo:hi], self.win_start[shard] - np.uint64(lo), self.win_len[shard])


def timed_steps(flt, batch, steps, warmup, barrier=None):
    """`warmup` untimed runs, then `steps` timed ones.  Returns (wall seconds, per-step sample-kernel ms, per-step ms of
    everything after it) -- the kernel times from HIP events on the kernels' own stream."""
    for _ in range(warmup):
        batch.run()
    flt.sync()
    if barrier:
        barrier()
    flt.profile_begin(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run()
    flt.sync()
    elapsed = time.perf_counter() - t0
    ms_sample, ms_rest = flt.profile_end(steps)
    return elapsed, ms_sample, ms_rest


def roofline_of(flt, inputs, rows_anded, n_rows, vote_ms):
    """The vote kernel against the HBM peak: algorithmic row bytes (rows ANDed x ceil(NB/8)) / kernel time."""
    algo = int(rows_anded) * inputs.row_bytes
    achieved = algo / (vote_ms * 1e-3) / 1e9
    index_bytes = (n_rows + 1) * flt.info()["row_pitch_bytes"]
    # FETCH_SIZE and the algorithmic count include reads the 256 MiB Infinity Cache serves.  A uniformly gathered table of
    # T bytes keeps about 256 MiB / T of itself there (MI355X_MICROARCH.md, Infinity Cache), so the HBM stacks themselves
    # move roughly (1 - share) of the bytes.  No counter sits behind that cache (TCC_EA0_RDREQ_DRAM counts requests
    # ADDRESSED to DRAM, cache hits included): the share is a model, and `roofline_large_index` is the measured answer.
    share = min(1.0, INFINITY_CACHE_BYTES / index_bytes)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None, "traffic_source": None, "kernel": "bmf_vote_kernel", "kernel_ms": vote_ms,
            "algorithmic_bytes_per_launch": algo, "index_bytes_in_hbm": int(index_bytes), "infinity_cache_share": share,
            "served_by": "HBM behind the 256 MiB Infinity Cache",
            "bound_note": f"hbm + infinity cache: about {share:.0%} of this index is cache-resident, so `achieved` can exceed what the HBM "
                          "stacks alone deliver (6.29 TB/s copy ceiling); `roofline_large_index` is the figure with 6 % cacheable",
            "hbm_side_estimate_GBps": achieved * (1.0 - share), "frac_hbm_side": achieved * (1.0 - share) / HBM_PEAK_GBPS,
            "note": "achieved = algorithmic row bytes / kernel time: it counts reads the Infinity Cache serves; "
                    "hbm_side_estimate_GBps = achieved x (1 - infinity_cache_share) is what the HBM stacks move (a model: "
                    "see roofline_large_index for the index the cache cannot help)"}


def candidate_checks(inputs, counts, buckets, shard):
    import numpy as np
    rd = inputs.reads
    n = shard.stop - shard.start
    strand = rd.truth_rc.astype(np.int64)[shard]
    idx = np.arange(n)
    own = buckets[idx, strand]                                   # candidate list on the true strand
    valid = np.arange(own.shape[1])[None, :] < counts[idx, strand][:, None]
    return {"reads_with_candidates": float((counts.sum(axis=1) > 0).mean()),
            "source_bucket_recovered": float(((own == rd.truth_bucket[shard][:, None]) & valid).any(axis=1).mean()),
            "candidates_per_read_per_strand": float(counts.mean())}


def pruned_leg(inputs, cli, device, k2i, steps, want, shard=None, all_reduce_max=None, barrier=None):
    """The same batch with BMF_FLAG_EARLY_EXIT: identical outputs from fewer row bytes.  Its "algorithmic
    bytes / time" would exceed the HBM peak because bytes are skipped, not moved faster, so it is reported beside the
    headline, never as `value`.  The first (untimed) run is the one on which the library measures which form to use."""
    import numpy as np
    import bucket_map_amd as bma
    fp = inputs.new_filter(cli, device, bma.BMF_FLAG_EARLY_EXIT, k2i)
    bp = inputs.batch(fp, shard)
    elapsed, ms_sample, ms_rest = timed_steps(fp, bp, steps, 2, barrier)
    if all_reduce_max:
        elapsed = all_reduce_max(elapsed)
    cp, bkp = bp.download()
    same = bool(np.array_equal(cp, want[0]))
    mask = np.arange(bkp.shape[-1])[None, None, :] < want[0][:, :, None]
    same = same and bool(np.array_equal(bkp[mask], want[1][mask]))
    info = fp.info()
    out = {"unit": "reads/s", "ms_per_step": elapsed / steps * 1e3, "kernels_ms_per_step": float(np.mean(ms_sample) + np.mean(ms_rest)),
           "outputs_identical_to_default_kernel": same, "outputs_identical_to_headline_run": same, "flag": "BMF_FLAG_EARLY_EXIT",
           "form": ("two passes" if info["pass1_rows"] else "one kernel"),
           "pass1_rows": info["pass1_rows"], "pass1_fold": info["pass1_fold"], "pass1_fold_rows": info["pass1_fold_rows"]}
    if info["pass1_rows"]:
        out["items_recounted"], out["items_slow_path"] = bp.pass2_counts()
        out["items"] = int(2 * len(cp))
        out["recount_column_loads"] = bp.recount_loads()
    return out, fp, bp


def candidate_pairs(counts, buckets):
    """The filter's candidate lists as the locator's (bucket, window, strand) triples, grouped by bucket as bucket_locator's
    loop meets them (bucket_locator.h:651-693: bucket by bucket; inside a bucket the read-as-is list, then the
    reverse-complement list, each in (read, window) order)."""
    import numpy as np
    mask = np.arange(buckets.shape[-1])[None, None, :] < counts[:, :, None]
    w_idx, s_idx, _ = np.nonzero(mask)
    b = buckets[mask]
    order = np.lexsort((w_idx, s_idx, b))
    return b[order].astype(np.uint32), w_idx[order].astype(np.uint32), s_idx[order].astype(np.uint8)


LOCATOR_SAMPLES, LOCATOR_MISMATCH_RATE, LOCATOR_INDEL_RATE = 10, 0.4, 0.02      # -p, -e, -n defaults (main.cpp:30-37)


def locator_leg(inp, cli, device, counts, buckets, steps, cpu_seconds, log, pmc=False):
    """north_star's second replaced subsystem, `bucket_map/locator`'s candidate scan: `_create_kmer_index` +
    `_find_offset` (bucket_locator.h:162-177,209-290) for EVERY candidate the filter produced for this batch, through
    bml_locate (include/bml.h).  Kernel times are HIP events inside the library (scan = bml_scan_kernel, replay = the light
    and heavy vote kernels); `ms_call` is the wall time of the C-ABI call with host buffers in and out.  Algorithmic bytes:
    every (bucket, chunk of candidates) reads its bucket once (bucket bases) and every k-mer occurrence is written once by
    the scan and read once by the replay (4 B each way: an occurrence is its offset in the bucket).  cpu_baseline: oracle/bm_locator_oracle.c::bmlo_locate, 1 thread,
    on the candidates of the first buckets (the analogue of the reference's per-bucket multimap build + vote)."""
    import numpy as np
    import bucket_map_amd as bma
    from bucket_map_amd import locate
    from oracle import oracle_c
    t_leg = time.perf_counter()
    k, p = cli["query_seed"], LOCATOR_SAMPLES
    L = bma.lib()
    mismatch = int(L.bmf_ceil_mul_f32(LOCATOR_MISMATCH_RATE, p))
    indel = int(L.bmf_ceil_mul_f32(LOCATOR_INDEL_RATE, inp.read_len))
    flat, _ = inp.genome.flat()
    bstart, blen = inp.genome.bucket_views(inp.bucket_len, inp.read_len)
    scan = locate.LocatorScan(k, p, mismatch, indel, inp.bucket_len + inp.read_len, device)
    scan.load_genome(flat, bstart, blen)
    rd = inp.reads
    minq = cli["average_base_quality"] * k
    t0 = time.perf_counter()
    sh, sp, has = scan.sample_windows(rd.bases, rd.quals, inp.win_start, inp.win_len, minq)
    sample_s = time.perf_counter() - t0
    pb, pw, pr = candidate_pairs(counts, buckets)
    keep = has[pw] != 0
    pb, pw, pr = pb[keep], pw[keep], pr[keep]
    seg = inp.win_len.astype(np.uint32)
    best = None
    runs = []
    for it in range(1 + max(1, steps)):                        # the first call sizes the occurrence buffer (untimed)
        t0 = time.perf_counter()
        off, votes = scan.locate(sh, sp, seg, pb, pw, pr)
        wall = time.perf_counter() - t0
        st = scan.stats()
        st["ms_call"] = wall * 1e3
        if it:
            runs.append(st)
    ms_scan = float(np.mean([r["ms_scan"] for r in runs]))
    ms_replay = float(np.mean([r["ms_replay"] for r in runs]))
    ms_host = float(np.mean([r["ms_host"] for r in runs]))
    st = runs[-1]
    chunk_max = (4096 // 2) // p                                # candidates per chunk (bml_kernels.hip.h: kTableSlots / 2 / p)
    run_len = np.diff(np.flatnonzero(np.concatenate(([True], pb[1:] != pb[:-1], [True]))))
    n_chunks = int(np.sum((run_len + chunk_max - 1) // chunk_max))
    bucket_bytes = int(np.sum(((run_len + chunk_max - 1) // chunk_max) * blen[pb[np.concatenate(([0], np.cumsum(run_len)[:-1]))]].astype(np.int64)))
    occ = int(st["occurrences"])
    algo = bucket_bytes + 8 * occ + int(pb.size) * (p * 10 + 9 + 8)   # + samples (hash u32, pos u16), group ends (u32) and the pair record in, (offset, votes) out
    kern_ms = ms_scan + ms_replay
    # the read's own position: a candidate on the true (bucket, strand) must be located at the simulated offset
    true_pair = (pb == rd.truth_bucket[pw]) & (pr == rd.truth_rc[pw])
    located = off > 0
    exact = true_pair & located & (np.abs(off.astype(np.int64) - rd.truth_offset[pw].astype(np.int64)) <= indel)
    leg = {"what": f"bml_locate over every candidate of the batch ({inp.profile} genome): -l {k} -p {p} mismatch {mismatch} indel {indel}, "
                   f"{int(pb.size)} candidates in {n_chunks} (bucket, chunk) workgroups",
           "unit": "candidates/s", "value": pb.size / (kern_ms * 1e-3), "candidates": int(pb.size), "chunks": n_chunks,
           "occurrences": occ, "heavy_candidates": int(st["heavy_candidates"]),
           "ms_scan": ms_scan, "ms_replay": ms_replay, "ms_host_between_kernels": ms_host,
           "ms_call": float(np.mean([r["ms_call"] for r in runs])),
           "ms_sample_windows_call": sample_s * 1e3, "timed_calls": len(runs),
           "roofline": {"bound": "hbm", "achieved": algo / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": "bml_scan_kernel + bml_replay_*",
                        "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": int(algo),
                        "bytes": {"bucket_bases": bucket_bytes, "occurrences_written_and_read": 8 * occ},
                        "scan_only": {"achieved": (bucket_bytes + 4 * occ) / (ms_scan * 1e-3) / 1e9,
                                      "frac": (bucket_bytes + 4 * occ) / (ms_scan * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                        "note": "the scan is a chain of dependent LDS reads per bucket k-mer (filter, table) and the vote an "
                                "order-dependent replay: the HBM fraction says how far the path is from moving its bytes at "
                                "stream speed, not that HBM is what it waits for (profiles/r04/locate_sq_counters_*.txt)"},
           "checks": {"candidates_located": float(located.mean()),
                      "true_candidates": int(true_pair.sum()),
                      "true_candidates_located_at_the_simulated_offset": float(exact.sum() / max(1, true_pair.sum()))}}
    # CPU baseline + parity: the oracle on the candidates of the first buckets, about `cpu_seconds` of one core
    if cpu_seconds > 0:
        ends = np.cumsum(run_len)
        prm = (k, p, mismatch, indel)
        n0 = int(ends[min(len(ends) - 1, 15)])
        t0 = time.perf_counter()
        o_ref, v_ref = oracle_c.locate(*prm, flat, bstart, blen, sh, sp, seg, pb[:n0], pw[:n0], pr[:n0])
        probe = time.perf_counter() - t0
        nb_want = int(min(len(ends), max(16, 16 * cpu_seconds / max(probe, 1e-3))))
        n1 = int(ends[nb_want - 1])
        t0 = time.perf_counter()
        o_ref, v_ref = oracle_c.locate(*prm, flat, bstart, blen, sh, sp, seg, pb[:n1], pw[:n1], pr[:n1])
        cpu_s = time.perf_counter() - t0
        leg["cpu_baseline"] = {"value": n1 / cpu_s, "unit": "candidates/s", "cores": 1, "kind": "port",
                               "sample": f"the {n1} candidates of the first {nb_want} candidate buckets, oracle/bm_locator_oracle.c "
                                         f"bmlo_locate -O3 (per bucket: index of all its k-mers, then the vote per candidate), "
                                         f"{cpu_s:.1f} s, {cpu_s / nb_want * 1e3:.2f} ms per bucket"}
        leg["checks"]["gpu_equals_oracle_on_sample"] = bool(np.array_equal(o_ref, off[:n1]) and np.array_equal(v_ref, votes[:n1]))
        leg["checks"]["parity_sample_candidates"] = n1
    hc, ho = scan.count_histogram(int(pb.size))
    leg["occurrences_per_candidate"] = {"bins": "[2^(b-1), 2^b), bin 0 = none", "candidates": [int(x) for x in hc[:24]],
                                        "occurrences": [int(x) for x in ho[:24]]}
    scan.close()
    # The scan's own bound on bases without repeats is its instruction count: a child under `rocprofv3 --pmc SQ_INSTS_VALU`
    # (uniform headline leg only; tools/bench_locate.py repeats this leg on its own) counts the scan kernel's VALU
    # wave-instructions, priced against the issue peak like the verifier's.
    if pmc and inp.profile == "uniform" and inp.workload == "egu":
        valu = live_pmc_counter([sys.executable, os.path.join(ROOT, "tools", "bench_locate.py"), "--genome-profile", "uniform", "--calls", "2",
                                 "--reads", str(rd.n)], "SQ_INSTS_VALU", "bml_scan_kernel", log)
        if valu:
            per = sorted(valu["values"])[len(valu["values"]) // 2]           # (the first call's scan may run twice: the median dispatch)
            ach = per / (ms_scan * 1e-3) / 1e9
            leg["roofline"]["scan_valu_issue"] = {
                "bound": "valu_issue", "achieved": ach, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instructions/s",
                "frac": ach / VALU_ISSUE_PEAK_GINST, "valu_wave_instructions_per_launch": per,
                "valu_wave_instructions_per_bucket_base": per / max(1, bucket_bytes),
                "source": f"live: rocprofv3 --pmc SQ_INSTS_VALU child of tools/bench_locate.py (bml_scan_kernel, median of "
                          f"{valu['dispatches']} dispatches); kernel time from the un-profiled run"}
    leg["seconds"] = time.perf_counter() - t_leg
    log(f"locator leg ({inp.profile}): {pb.size} candidates, {occ} occurrences, scan {ms_scan:.2f} ms + replay {ms_replay:.2f} ms "
        f"(call {leg['ms_call']:.1f} ms), {leg['roofline']['frac']:.3f} of the HBM peak, {leg['seconds']:.0f} s")
    return leg


def skewed_leg(args, device, cli, k2i, threads, log):
    """BASELINE configs[1]'s geometry on the genome-LIKE genome (bm_synth.h): default kernel, pruned kernels, parity sample."""
    import numpy as np
    from oracle import oracle_c
    total_bp, bucket_len, read_len, n_reads = WORKLOADS[args.workload]
    n_reads = args.reads or n_reads
    t0 = time.perf_counter()
    inp = Inputs(args.workload, args.total_bp or total_bp, args.bucket_len or bucket_len, read_len, n_reads, "genome", threads)
    flt = inp.new_filter(cli, device, 0, k2i)
    batch = inp.batch(flt)
    elapsed, ms_sample, ms_vote = timed_steps(flt, batch, args.steps, 1)
    counts, buckets = batch.download()
    rows_anded = batch.rows_anded()
    vote_ms = float(np.mean(ms_vote))
    n_rows = int((k2i >= 0).sum())
    zeros = flt.zeros()
    thr = int(np.float32(cli["distinguishability"]) * np.float32(inp.nb))
    roof = roofline_of(flt, inp, rows_anded, n_rows, vote_ms)
    for k in ("note", "traffic", "traffic_source"):
        roof.pop(k)
    leg = {"what": f"{args.workload}-like geometry on the genome-like synthetic genome (bm_synth.h: order-6 Markov base layer, repeat "
                   f"families, satellites, segmental duplications, gaps), NB={inp.nb}, {n_reads} x {read_len} bp reads per step, "
                   f"params {args.params}",
           "value": n_reads * args.steps / elapsed, "unit": "reads/s", "ms_per_step": elapsed / args.steps * 1e3,
           "rows_passing_distinguishability": float((zeros >= thr).mean()),
           "rows_passing_reference_log": "95.8 % on GRCh38 at bucket_len 65 536 (bucket_map/benchmark/short_read/log/bucketmap_3_map.log:8)",
           "roofline": roof, "sample_kernel_ms": float(np.mean(ms_sample)),
           "bytes_per_read": (rows_anded * inp.row_bytes + 2 * int(inp.win_len.sum())) / n_reads,
           "checks": candidate_checks(inp, counts, buckets, slice(0, n_reads))}
    leg["checks"]["reads_with_candidates_reference_log"] = "94.9 % (bucketmap_3_map.log:12)"
    # parity on a sample: the oracle over all host threads (a checker here, not a baseline)
    n_cpu = min(args.skewed_parity_reads, n_reads)
    if n_cpu:
        from concurrent.futures import ThreadPoolExecutor
        ora = oracle_c.Index(oracle_c.params_from_cli(inp.nb, **cli), flt.index_download(), k2i)
        n_thr = usable_cores()
        cuts = [n_cpu * t // n_thr for t in range(n_thr + 1)]
        with ThreadPoolExecutor(n_thr) as pool:
            parts = list(pool.map(lambda t: ora.map_windows(inp.reads.bases, inp.reads.quals, inp.win_start[cuts[t]:cuts[t + 1]],
                                                            inp.win_len[cuts[t]:cuts[t + 1]]), range(n_thr)))
        c_ref = np.concatenate([p[0] for p in parts])
        b_ref = np.concatenate([p[1] for p in parts])
        mask = np.arange(b_ref.shape[-1])[None, None, :] < c_ref[:, :, None]
        leg["checks"]["gpu_equals_oracle_on_sample"] = bool(np.array_equal(c_ref, counts[:n_cpu]) and
                                                            np.array_equal(b_ref[mask], buckets[:n_cpu][mask]))
        leg["checks"]["parity_sample_reads"] = int(n_cpu)
        del ora
    batch.close()
    flt.close()
    pruned, fp, bp = pruned_leg(inp, cli, device, k2i, args.steps, (counts, buckets))
    pruned["value"] = n_reads / (pruned["ms_per_step"] * 1e-3)
    bp.close()
    fp.close()
    leg["pruned"] = pruned
    if not args.no_locator_leg:
        try:
            leg["locator"] = locator_leg(inp, cli, device, counts, buckets, min(args.steps, 3), args.locator_cpu_seconds, log)
        except Exception as e:                                 # an optional leg must not cost the others
            leg["locator"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    leg["seconds"] = time.perf_counter() - t0
    log(f"skewed leg: vote {vote_ms:.2f} ms ({leg['roofline']['frac']:.3f} of peak), pruned {pruned['ms_per_step']:.2f} ms/step "
        f"(fold {pruned['pass1_fold']} x {pruned['pass1_fold_rows']} rows), {leg['seconds']:.0f} s")
    return leg


def large_index_leg(args, device, k2i, threads, log):
    """The vote kernel on an index the Infinity Cache cannot help: 2.29 Gbp at bucket_len 16 384 (NB = 140 471, 4.63 GB)."""
    import numpy as np
    t0 = time.perf_counter()
    read_len = 300
    inp = Inputs("egu", LARGE_INDEX["total_bp"], LARGE_INDEX["bucket_len"], read_len, args.large_index_reads, "uniform", threads)
    flt = inp.new_filter(cli_params("default", read_len), device, 0, k2i)
    batch = inp.batch(flt)
    steps = max(2, min(args.steps, 3))
    elapsed, ms_sample, ms_vote = timed_steps(flt, batch, steps, 1)
    counts, buckets = batch.download()
    n_rows = int((k2i >= 0).sum())
    roof = roofline_of(flt, inp, batch.rows_anded(), n_rows, float(np.mean(ms_vote)))
    for k in ("note", "traffic", "traffic_source"):
        roof.pop(k)
    leg = {"what": f"uniform synthetic genome {inp.genome.total_length()} bp at bucket_len {inp.bucket_len}: NB={inp.nb} "
                   f"(one wave per 65 536-bucket slice + merge), {n_rows} rows x {inp.row_bytes} B, {inp.reads.n} x {read_len} bp reads, "
                   f"params default",
           "value": inp.reads.n * steps / elapsed, "unit": "reads/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
           **roof, "checks": candidate_checks(inp, counts, buckets, slice(0, inp.reads.n))}
    batch.close()
    flt.close()
    leg["seconds"] = time.perf_counter() - t0
    log(f"large-index leg: vote {leg['kernel_ms']:.1f} ms = {leg['achieved']:.0f} GB/s ({leg['frac']:.3f} of peak, "
        f"{leg['infinity_cache_share']:.3f} of the index cacheable), {leg['seconds']:.0f} s")
    return leg


VALU_ISSUE_PEAK_GINST = 256 * 4 * 2.4 / 2      # wave64 VALU instructions/s, in G: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per
                                               # instruction (MI355X_MICROARCH.md: "issues each VALU instruction over 2 cycles",
                                               # table row `v_fma_f32` (wave64) 2 cyc; max clock 2400 MHz) = 1 228.8


def verifier_leg(log, pmc=True):
    """The alignment verifier (include/bmv.h) beside the headline: tools/bench_verify.py in a process of
    its own, on the short-read shapes and the long-read shape DESIGN.md 4.4 quotes.  No oracle in that process
    (--cpu-sample 0): GPU == oracle is tests/test_align.py's business.  The kernels are integer VALU work (Myers bit-vector
    columns), so the bound is the chip's vector issue rate: each shape runs a second time under `rocprofv3 --pmc
    SQ_INSTS_VALU` (a pass of its own, nothing traced) and `roofline` = counted wave-instructions / kernel time against
    256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles."""
    import subprocess
    t0 = time.perf_counter()
    leg = {"what": "bmv_align: semi-global edit distance + CIGAR of every candidate alignment (Myers bit-vector, checkpoints, "
                   "traceback on the device); kernel time from HIP events inside the library, inputs resident",
           "unit": "cell updates/s"}
    tool = os.path.join(ROOT, "tools", "bench_verify.py")
    for name, extra in (("short_reads", ["--reads", "1000000", "--len", "300"]),
                        ("short_reads_150", ["--reads", "1000000", "--len", "150"]),
                        ("long_reads", ["--reads", "20000", "--len", "10000", "--indel-rate", "0.1", "--sub", "0.03"])):
        try:
            cmd = [sys.executable, tool, "--cpu-sample", "0", "--repeat", "2"] + extra
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, check=True).stdout
            d = json.loads(out.strip().splitlines()[-1])
            leg[name] = {"alignments": d["config"]["alignments"], "query_len": d["config"]["query_len"], "text_len": d["config"]["text_len"],
                         "ms_kernels": d["ms_kernels"], "value": d["cell_updates_per_s"], "alignments_per_s": d["value"],
                         "mean_edits": d["mean_edits"]}
            cells = d["cell_updates_per_s"] * d["ms_kernels"] * 1e-3
            valu = live_pmc_counter(cmd, "SQ_INSTS_VALU", "bmv_align", log) if pmc else None
            if valu:
                per_call = valu["sum"] / 2                                # --repeat 2: two identical calls
                ach = per_call / (d["ms_kernels"] * 1e-3) / 1e9
                leg[name]["roofline"] = {
                    "bound": "valu_issue", "achieved": ach, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instructions/s",
                    "frac": ach / VALU_ISSUE_PEAK_GINST, "valu_wave_instructions_per_call": per_call,
                    "valu_wave_instructions_per_cell": per_call / cells, "cells_per_call": cells,
                    "cycles_per_instruction_and_simd": 256 * 4 * 2.4e9 * d["ms_kernels"] * 1e-3 / per_call,
                    "source": f"live: rocprofv3 --pmc SQ_INSTS_VALU child of tools/bench_verify.py, bmv_align* kernels, "
                              f"{valu['dispatches']} dispatches / 2 calls; kernel time from the un-profiled run",
                    "note": "wave64 on SIMD-32: 2 cycles per VALU instruction is the issue peak; this kernel's 64-bit integer "
                            "shifts, adds with carry and v_alignbit sustain about 4 (DESIGN.md 4.4)"}
        except (subprocess.SubprocessError, ValueError, KeyError, IndexError) as e:
            leg[name] = {"error": str(e)[:300]}
    leg["seconds"] = time.perf_counter() - t0
    log("verifier leg: " + ", ".join(f"{k} {v['value'] / 1e12:.1f} T cell updates/s ({v['ms_kernels']:.2f} ms"
                                     + (f", {v['roofline']['frac']:.2f} of VALU issue" if "roofline" in v else "") + ")"
                                     for k, v in leg.items() if isinstance(v, dict) and "value" in v) + f", {leg['seconds']:.0f} s")
    return leg


def live_pmc_counter(cmd, counter, kernel_substr, log):
    """Runs `cmd` under `rocprofv3 --pmc <counter>` (a pass of its own: no tracing) and returns {"sum", "dispatches", "values"}
    of the counter over the kernels whose name contains `kernel_substr`; None when the profiler is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if under_profiler():
        return None
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="bm_pmc_", dir="/tmp")
    try:
        r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + cmd, cwd="/tmp",
                           env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=600)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            log(f"pmc child ({counter}) failed (rc {r.returncode}): {r.stderr[-300:]}")
            return None
        vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(max(files, key=os.path.getmtime)))
                if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]]
        return {"sum": sum(vals), "dispatches": len(vals), "values": vals} if vals else None
    except (OSError, ValueError, KeyError, subprocess.SubprocessError) as e:
        log(f"pmc child ({counter}) failed: {e}")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def under_profiler():
    """True when this process already runs under rocprofv3 / rocprofiler-sdk (a nested profiler must not be started)."""
    pre = os.environ.get("LD_PRELOAD", "")
    return any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in os.environ) or "rocprof" in pre


def live_pmc_traffic(args, log):
    """Runs this file again as `--pmc-child` under rocprofv3 --pmc FETCH_SIZE and returns the roofline fields
    {traffic, traffic_source, traffic_dispatches, traffic_over_algorithmic}, or None when the profiler is not usable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if under_profiler():
        log("already under a profiler: no nested rocprofv3 child")
        return None
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="bm_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
           "--pmc-child", "--workload", args.workload, "--params", args.params, "--kmer-frac", str(args.kmer_frac),
           "--reads", str(args.reads), "--total-bp", str(args.total_bp), "--bucket-len", str(args.bucket_len),
           "--host-threads", str(args.host_threads), "--genome-profile", args.genome_profile]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=600)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            log(f"pmc child failed (rc {r.returncode}): {r.stderr[-400:]}")
            return None
        child = json.loads(r.stdout.strip().splitlines()[-1])
        vals = []
        for row in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = row["Kernel_Name"]
            if row["Counter_Name"] == "FETCH_SIZE" and "bmf_vote_kernel" in k:
                vals.append(float(row["Counter_Value"]))
        if not vals:
            return None
        traffic = sum(vals) / len(vals) * 1024 * 2
        log(f"pmc child: {len(vals)} vote dispatches, FETCH_SIZE mean {sum(vals) / len(vals):.0f} KiB ({time.perf_counter() - t0:.0f}s)")
        return {"traffic": int(traffic), "traffic_dispatches_KiB": vals,
                "traffic_over_algorithmic": traffic / child["algorithmic_bytes_per_launch"],
                "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE child run of this bench (same workload, vote kernel, "
                                  f"mean of {len(vals)} dispatches) x1024 (KiB) x2 (gfx950 wide streaming reads; the factor is "
                                  "checked against the request-size counters in profiles/r03/fetch_size_crosscheck.txt)"}
    except (OSError, ValueError, KeyError, subprocess.SubprocessError) as e:
        log(f"pmc child failed: {e}")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pmc_child(args):
    """The profiled child of live_pmc_traffic: same synthetic workload, three launches of the hot path, nothing else
    (no torch, no CPU leg).  Prints the algorithmic bytes of one vote launch."""
    from bucket_map_amd import host
    total_bp, bucket_len, read_len, n_reads = WORKLOADS[args.workload]
    threads = args.host_threads or usable_cores()
    cli = cli_params(args.params, read_len)
    inp = Inputs(args.workload, args.total_bp or total_bp, args.bucket_len or bucket_len, read_len, args.reads or n_reads,
                 args.genome_profile, threads)
    flt = inp.new_filter(cli, 0, 0, host.select_qgrams(cli["index_seed"], args.kmer_frac))
    batch = inp.batch(flt)
    for _ in range(3):
        batch.run()
    flt.sync()
    print(json.dumps({"algorithmic_bytes_per_launch": int(batch.rows_anded()) * inp.row_bytes}), flush=True)
    batch.close()
    flt.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="egu", choices=sorted(WORKLOADS) + [w + "-skewed" for w in sorted(WORKLOADS)],
                    help="<name>-skewed: the same geometry on the genome-like genome as the headline (= --genome-profile genome)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the workload's)")
    ap.add_argument("--total-bp", type=int, default=0, help="override the workload's genome size (other NB geometries)")
    ap.add_argument("--bucket-len", type=int, default=0, help="override the workload's bucket length (other NB geometries)")
    ap.add_argument("--params", default="default", choices=["default", "bench"],
                    help="default = CLI defaults (k12 q9 S15 F6); bench = benchmark_map.sh (-s 20 -e 0.6 -l 14 -b 10)")
    ap.add_argument("--cpu-sample", type=int, default=200000, help="reads timed on the CPU oracle (0 = skip)")
    ap.add_argument("--kmer-frac", type=float, default=1.0,
                    help="-f of the index (FracMinHash, seeded): 1 = all 4^q rows (the roofline configuration); 0.25 = the "
                         "reference's default, a 217 MB index that fits the 256 MiB Infinity Cache (secondary data point)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--genome-profile", default="uniform", choices=["uniform", "genome"],
                    help="the HEADLINE's genome: uniform = i.i.d. bases (SURVEY 8d); genome = bm_synth.h's skewed, repetitive generator")
    ap.add_argument("--index-build", default="gpu", choices=["gpu", "host"],
                    help="where the synthetic index is built (setup only, outside the timed region)")
    ap.add_argument("--early-exit", action="store_true",
                    help="BMF_FLAG_EARLY_EXIT: identical outputs, fewer rows actually read (off by default so "
                         "that the roofline line prices exactly the reference's row reads)")
    ap.add_argument("--no-pruned-leg", action="store_true", help="skip the extra BMF_FLAG_EARLY_EXIT measurement")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the `skewed`, `roofline_large_index` and `verifier` legs (N = 1 only anyway)")
    ap.add_argument("--no-locator-leg", action="store_true", help="skip the `locator` legs (bml_locate over the batch's candidates)")
    ap.add_argument("--locator-cpu-seconds", type=float, default=8.0, help="CPU time of the locator oracle sample per leg (0 = skip)")
    ap.add_argument("--skewed-parity-reads", type=int, default=50000, help="reads of the skewed leg checked against the oracle")
    ap.add_argument("--large-index-reads", type=int, default=500000, help="reads per step of the large-index leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--device-override", type=int, default=-1,
                    help="rehearsal only: put every rank on this device (use with --backend gloo)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="N > 1: strong = the workload's reads cut into N contiguous shards (BASELINE configs[2], the "
                         "default); weak = every rank maps its own full batch")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not spawn the rocprofv3 --pmc FETCH_SIZE child that measures roofline.traffic live")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.workload.endswith("-skewed"):        # `--workload egu-skewed` = `--workload egu --genome-profile genome`
        args.workload, args.genome_profile = args.workload[: -len("-skewed")], "genome"
    if args.pmc_child:
        return pmc_child(args)

    t_process = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # torch first: its bundled HIP runtime (soname libamdhip64.so.7) must be the one libbmf.so binds to,
    # otherwise the process would hold two HIP runtimes.
    import torch
    import torch.distributed as dist
    import numpy as np

    device = local_rank if args.device_override < 0 else args.device_override
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend="gloo")
    comm_dev = "cuda" if (world > 1 and args.backend == "nccl") else "cpu"

    import bucket_map_amd as bma
    from bucket_map_amd import host

    total_bp, bucket_len, read_len, n_reads = WORKLOADS[args.workload]
    n_reads = args.reads or n_reads
    threads = args.host_threads or max(1, usable_cores() // world)
    cli = cli_params(args.params, read_len)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---------------- synthetic inputs, identical on every rank except the reads
    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    if world == 1:
        scaling = "weak"          # one GPU: the two coincide; the contract's default label
    # strong: every rank simulates the same batch (same seed) and keeps the windows of its contiguous shard
    inp = Inputs(args.workload, args.total_bp or total_bp, args.bucket_len or bucket_len, read_len, n_reads, args.genome_profile,
                 threads, read_seed=20240003 + (7919 * rank if scaling == "weak" else 0))
    genome, reads, nb, row_bytes = inp.genome, inp.reads, inp.nb, inp.row_bytes
    log(f"genome ({args.genome_profile}): {len(inp.lens)} records, {genome.total_length()} bp, NB={nb} ({inp.genome_s:.1f}s); "
        f"reads: {reads.n} x {read_len} bp ({inp.reads_s:.1f}s), scaling {scaling}")

    # ---------------- GPU side
    t0 = time.perf_counter()
    k2i = host.select_qgrams(cli["index_seed"], args.kmer_frac)
    n_rows = int((k2i >= 0).sum())
    index = None
    flags = bma.BMF_FLAG_EARLY_EXIT if args.early_exit else 0
    if args.index_build == "host":
        index = host.Index(genome, nb, inp.bucket_len, read_len, q=cli["index_seed"], kmer_frac=args.kmer_frac, threads=threads)
        flt = bma.Filter(bma.Params.from_cli(nb, device=device, flags=flags, **cli))
        flt.load_index_ptr(index.rows_ptr, index.num_rows, index.k2i_ptr, index.num_kmers)
    else:
        flt = inp.new_filter(cli, device, flags, k2i)
    params = flt.params
    log(f"index in HBM via {args.index_build} build: {n_rows} rows x {row_bytes} B = {n_rows * row_bytes / 1e6:.1f} MB "
        f"({time.perf_counter() - t0:.1f}s), kernel variant {flt.info()}")
    shard = slice(0, reads.n)
    if scaling == "strong" and world > 1:
        shard = slice(reads.n * rank // world, reads.n * (rank + 1) // world)
    batch = inp.batch(flt, shard)
    n_mine = shard.stop - shard.start
    reads_per_step = reads.n if scaling == "strong" else world * reads.n
    setup_s = time.perf_counter() - t_process

    for _ in range(args.warmup):
        batch.run()
    flt.sync()
    barrier()
    flt.profile_begin(args.steps)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
    flt.sync()
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t_start
    ms_sample, ms_vote = flt.profile_end(args.steps)
    elapsed = all_reduce_max(my_elapsed)
    if world > 1:
        dist.barrier()

    rows_anded = batch.rows_anded()
    counts, buckets = batch.download()

    # per-rank view of the timed region (N > 1): the slowest rank sets `value`; whether the others were waiting for it
    # (jitter) or every rank lost the same time (a serial part, launch gaps) is read off these.
    per_rank = None
    if world > 1:
        kern = ms_sample + ms_vote                                        # per step, HIP events
        mine = torch.tensor([my_elapsed / args.steps * 1e3, float(kern.mean()), float(kern.min()), float(kern.max()),
                             setup_s, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, float(n_mine)],
                            dtype=torch.float64, device=comm_dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        rows = np.array([g.cpu().numpy() for g in got])
        per_rank = {"step_ms": [round(float(x), 3) for x in rows[:, 0]],
                    "kernel_ms_mean": [round(float(x), 3) for x in rows[:, 1]],
                    "kernel_ms_min": [round(float(x), 3) for x in rows[:, 2]], "kernel_ms_max": [round(float(x), 3) for x in rows[:, 3]],
                    "launch_gap_ms": [round(float(a - b), 3) for a, b in zip(rows[:, 0], rows[:, 1])],
                    "setup_wall_s": [round(float(x), 1) for x in rows[:, 4]], "max_rss_MB": [round(float(x)) for x in rows[:, 5]],
                    "reads": [int(x) for x in rows[:, 6]],
                    "what": "step_ms = wall per step on that rank; kernel_ms_* = sample + vote kernels per step (HIP events); "
                            "launch_gap_ms = step_ms - kernel_ms_mean; setup = process start to first warm-up step"}

    # the other scaling mode, beside the headline (strong runs only): every rank maps the whole batch
    weak_leg = None
    if scaling == "strong" and world > 1:
        wb = inp.batch(flt)
        w_elapsed, _, _ = timed_steps(flt, wb, args.steps, 1, barrier)
        weak_s = all_reduce_max(w_elapsed)
        wb.close()
        weak_leg = {"value": world * reads.n * args.steps / weak_s, "unit": "reads/s", "ms_per_step": weak_s / args.steps * 1e3,
                    "what": f"every rank maps its own copy of the {reads.n}-read batch (per-GPU work fixed)"}

    # PCIe-inclusive rate of the host-buffer entry point (bmf_map_windows: H2D of the reads, both kernels,
    # compacted D2H of the results; the batch goes through in pieces so that the copies of one piece run under
    # the kernels of its neighbours).  From page-locked buffers and from ordinary pageable memory.  Reported for
    # DESIGN.md only; it is never `value`.
    out_arrays = (np.zeros((len(inp.win_start), 2), np.uint32), np.empty(2 * len(inp.win_start) * params.max_candidates, np.uint32))

    def time_map_windows(f, b, q):
        f.map_windows_compact(b, q, inp.win_start, inp.win_len, out=out_arrays)          # first call allocates
        best = 1e9
        for _ in range(3):
            t_h = time.perf_counter()
            f.map_windows_compact(b, q, inp.win_start, inp.win_len, out=out_arrays)
            best = min(best, time.perf_counter() - t_h)
        return best
    host_pinned_s = host_buffer_s = None
    pinned_b = pinned_q = None
    if world == 1:
        pinned_b, pinned_q = bma.pinned_copy(reads.bases), bma.pinned_copy(reads.quals)
        host_pinned_s = time_map_windows(flt, pinned_b.array, pinned_q.array)
        host_buffer_s = time_map_windows(flt, reads.bases, reads.quals)

    # Second leg, reported beside the headline, never instead of it: the same batch with BMF_FLAG_EARLY_EXIT.
    pruned = None
    if not args.early_exit and not args.no_pruned_leg:
        pruned, fp, bp = pruned_leg(inp, cli, device, k2i, args.steps, (counts, buckets), shard, all_reduce_max,
                                    (dist.barrier if world > 1 else None))
        pruned["value"] = reads_per_step / (pruned["ms_per_step"] * 1e-3)
        bp.close()
        if world == 1:
            pruned["pcie_inclusive_ms_pinned"] = time_map_windows(fp, pinned_b.array, pinned_q.array) * 1e3
        fp.close()
    pinned_b = pinned_q = None

    result = None
    if rank == 0:
        reads_per_s = reads_per_step * args.steps / elapsed
        vote_ms = float(np.mean(ms_vote))
        roof = roofline_of(flt, inp, rows_anded, n_rows, vote_ms)
        roof["bytes_per_read"] = (roof["algorithmic_bytes_per_launch"] + 2 * int(inp.win_len[shard].sum())) / n_mine
        roof["sample_kernel_ms"] = float(np.mean(ms_sample))
        result = {
            "metric": "mapped reads/sec (1M×300bp, 65536-bucket index)",   # BASELINE.json's metric, its first clause
            "value": reads_per_s, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": f"{args.workload}-like synthetic genome ({args.genome_profile} profile) {genome.total_length()} bp, bucket_len {inp.bucket_len}, "
                            f"NB={nb}, -f {args.kmer_frac:g} index ({n_rows} rows x {row_bytes} B), {reads_per_step} x {read_len} bp "
                            f"simulated reads per step (sub 0.002, ins=del 0.00025), params {args.params} "
                            f"(k={params.k} q={params.q} S={params.num_samples} F={params.num_fault})",
                "reads_per_gpu": int(n_mine), "global_reads_per_step": int(reads_per_step),
                "parallelism": f"reads sharded over {world} GPU(s) in contiguous ranges, index replicated, no collective",
                "early_exit": bool(args.early_exit),
            },
            "roofline": roof,
            "checks": candidate_checks(inp, counts, buckets, shard),
            "pruned": pruned,
            "weak_scaling": weak_leg,
            "per_rank": per_rank,
            "setup": {"wall_s": setup_s, "genome_s": inp.genome_s, "reads_s": inp.reads_s, "host_threads": threads,
                      "max_rss_MB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0},
        }
        if host_pinned_s is not None:
            result["pcie_inclusive"] = {"reads_per_s_per_gpu": reads.n / host_pinned_s, "ms": host_pinned_s * 1e3,
                                        "ms_pageable": host_buffer_s * 1e3, "reads": int(reads.n),
                                        "what": "bmf_map_windows_compact (the entry point the bucketmap tool calls), host buffers in and "
                                                "out: H2D of the reads, kernels and packed D2H pipelined in pieces; page-locked source "
                                                "(ms) and pageable source (ms_pageable)"}

        # ---------------- CPU baseline (oracle = port of the reference algorithm, 1 thread) + parity sample
        if args.cpu_sample > 0 and world == 1:               # the CPU baseline is an N=1 leg only
            from oracle import oracle_c
            n_cpu = min(args.cpu_sample, reads.n)
            rows_host = index.rows() if index is not None else flt.index_download()
            ora = oracle_c.Index(oracle_c.params_from_cli(nb, **cli), rows_host, k2i)
            t0 = time.perf_counter()
            c_ref, b_ref, rows_ref = ora.map_windows(reads.bases, reads.quals, inp.win_start[:n_cpu], inp.win_len[:n_cpu])
            cpu_s = time.perf_counter() - t0
            same = bool(np.array_equal(c_ref, counts[:n_cpu]))
            mask = np.arange(b_ref.shape[-1])[None, None, :] < c_ref[:, :, None]
            same = same and bool(np.array_equal(b_ref[mask], buckets[:n_cpu][mask]))
            result["cpu_baseline"] = {"value": n_cpu / cpu_s, "unit": "reads/s", "cores": 1, "kind": "port",
                                      "sample": f"first {n_cpu} reads of the same batch, oracle/bm_oracle.c -O3, "
                                                f"{cpu_s:.1f} s, {cpu_s / n_cpu * 1e6:.1f} us/read"}
            result["checks"]["gpu_equals_oracle_on_sample"] = same
            result["checks"]["parity_sample_reads"] = int(n_cpu)
            # the same port on ALL host cores this process may use (reads sharded over threads; the
            # oracle is re-entrant and ctypes releases the GIL), as SURVEY.md 8d asks beside the 1-thread figure
            from concurrent.futures import ThreadPoolExecutor
            n_thr = usable_cores()
            per = max(1, min(reads.n // n_thr, 40000))
            shards = [(i * per, (i + 1) * per) for i in range(n_thr)]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(n_thr) as pool:
                list(pool.map(lambda s: ora.map_windows(reads.bases, reads.quals, inp.win_start[s[0]:s[1]], inp.win_len[s[0]:s[1]]),
                              shards))
            mt_s = time.perf_counter() - t0
            result["cpu_baseline_all_cores"] = {"value": n_thr * per / mt_s, "unit": "reads/s", "cores": n_thr,
                                                "kind": "port", "sample": f"{n_thr} threads x {per} reads, {mt_s:.1f} s"}
            del ora, rows_host
        else:
            result["cpu_baseline"] = None

    # every GPU resource of the headline is released before the other legs and the profiler child start
    batch.close()
    flt.close()
    if rank == 0 and world == 1 and not args.no_locator_leg and not args.no_extra_legs:
        try:
            result["locator"] = locator_leg(inp, cli, device, counts, buckets, min(args.steps, 3), args.locator_cpu_seconds, log,
                                            pmc=not args.no_pmc and not args.reads and not args.total_bp and not args.bucket_len)
        except Exception as e:                                 # an optional leg must not cost the headline record
            result["locator"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    del inp, reads, genome, index

    if rank == 0:
        extra = world == 1 and not args.no_extra_legs and not args.early_exit and args.kmer_frac == 1.0
        # HBM-side traffic of the vote kernel, measured NOW: PMC counters cannot be read in-process, so a child
        # runs the same workload's vote kernel under `rocprofv3 --pmc FETCH_SIZE` (its own pass, no tracing) and the
        # per-dispatch values are corrected as MI355X_MICROARCH.md prescribes (KiB -> bytes, x2 on gfx950 for wide
        # streaming reads).  If the profiler cannot run here the committed pass of the same workload is quoted,
        # and traffic_source says so.
        if world == 1 and not args.no_pmc and not args.early_exit:
            live = live_pmc_traffic(args, log)
            if live:
                result["roofline"].update(live)
        if result["roofline"]["traffic"] is None:
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
                    pmc = json.load(f)
                for e in pmc.get("entries", [pmc]):
                    if (e.get("workload") == args.workload and e.get("params") == args.params
                            and e.get("reads") == int(n_mine) and not args.early_exit and args.kmer_frac == 1.0
                            and not args.total_bp and not args.bucket_len and args.genome_profile == "uniform"):
                        result["roofline"]["traffic"] = e["vote_kernel_traffic_bytes"]
                        result["roofline"]["traffic_source"] = "NOT measured in this run; committed pass " + e["source"]
            except (OSError, ValueError, KeyError):
                pass
        def guarded(name, fn, *a):
            """An optional leg: its failure (out of memory on a shared card, an import error) is recorded under its key and
            never costs the headline, roofline and cpu_baseline already measured."""
            try:
                result[name] = fn(*a)
            except Exception as e:
                result[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
                log(f"{name} leg failed: {result[name]['error']}")
        if extra and args.genome_profile == "uniform":
            guarded("skewed", skewed_leg, args, device, cli, k2i, threads, log)
        if extra and args.workload == "egu" and not args.total_bp and not args.bucket_len:
            guarded("roofline_large_index", large_index_leg, args, device, k2i, threads, log)
        if extra and args.workload == "egu" and not args.total_bp and not args.bucket_len:
            guarded("verifier", verifier_leg, log, not args.no_pmc)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
