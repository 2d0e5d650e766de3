#!/usr/bin/env python3
"""End-to-end run of the `bucketmap` tool on a synthetic genome (GPU box): writes FASTA + FASTQ, indexes,
maps with the GPU mapper + GPU locator scan, reports the tool's own [BENCHMARK] lines and the accuracy
against the simulator's ground truth.

    python tools/e2e_cli.py --workload egu --reads 1000000 --out gpurun_out/e2e_egu.txt
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bucket-map_amd", "python"))
import bench  # noqa: E402
from bucket_map_amd import host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="egu")
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--dir", default="/tmp/bm_e2e")
ap.add_argument("--out", default="")
ap.add_argument("--extra", default="", help="extra CLI flags, e.g. '--early-exit'")
ap.add_argument("--align", action="store_true", help="run bucketmap_align (alignment verification + CIGAR)")
ap.add_argument("--kmer-frac", default="1", help="-f of the index (the reference's default is 0.25)")
ap.add_argument("--bucket-len", type=int, default=0, help="override the workload's bucket length (BASELINE configs[4]: 262144)")
ap.add_argument("--index-seed", default="", help="-k of the index (default 9; 262144-bp buckets of a uniform genome need 10)")
ap.add_argument("--gpus", default="", help="--gpus of the tool, e.g. 0,0,0")
ap.add_argument("--profile", default="uniform", choices=["uniform", "genome"],
                help="uniform = i.i.d. bases; genome = bm_synth.h's skewed, repetitive genome (q = 9 works at 262144-bp buckets)")
ap.add_argument("--verbose", action="store_true", help="BML_LOG / BMV_LOG_CLASSES: the locator's groups and the verifier's rounds")
ap.add_argument("--long", action="store_true",
                help="the reference's long-read profile (benchmark/long_read/benchmark_map.sh:25): 10-kbp ONT-like reads "
                     "(sub 0.03, ins = del 0.025), -s 30 -e 0.9 -n 0.1 -l 12 -p 20 -u 5; the workload's 65536-bp buckets "
                     "are kept: on a uniform random genome 262144-bp buckets make every 9-mer row 63 % dense, no q-gram "
                     "passes -d 0.5 and nothing maps -- real genomes are not uniform")
args = ap.parse_args()

total_bp, bucket_len, read_len, _ = bench.WORKLOADS[args.workload]
bucket_len = args.bucket_len or bucket_len
if args.long:
    read_len = 300              # the long-read script keeps the default -r 300: reads are cut into 5 windows of 300
sim_len, err, profile = read_len, {}, []
if args.long:
    sim_len = 10000
    err = dict(sub=0.03, ins=0.025, dele=0.025)
    profile = ["-s", "30", "-e", "0.9", "-n", "0.1", "-l", "12", "-p", "20", "-u", "5"]
os.makedirs(args.dir, exist_ok=True)
log = []


def say(msg):
    print(msg, flush=True)
    log.append(msg)


t = time.perf_counter()
lens = [total_bp] if args.workload == "mini" else bench.workload_record_lengths(args.workload, total_bp)
g = host.Genome.synth(20240001, lens, 16, profile=args.profile)
g.write_fasta(os.path.join(args.dir, "g.fa"))
rd = host.Reads(g, bucket_len, read_len, sim_len, args.reads, seed=20240003, threads=16, **err)
rd.write_fastq(os.path.join(args.dir, "reads"))
say(f"[e2e] inputs written in {time.perf_counter() - t:.1f} s ({g.total_length()} bp, {rd.n} reads)")
exe = os.path.join(ROOT, "bucket-map_amd", "bucketmap")
common = ["-i", "idx", "--genome", "g.fa", "--bucket-len", str(bucket_len), "-r", str(read_len), "-f", args.kmer_frac, *profile]
if args.index_seed:
    common += ["-k", args.index_seed]
if args.gpus:
    common += ["--gpus", args.gpus]
for f in ("idx.qgram", "idx.kmers_index", "idx.bucket_id", "out.sam"):
    p = os.path.join(args.dir, f)
    if os.path.exists(p):
        os.remove(p)
t = time.perf_counter()
r = subprocess.run([exe, "-x", *common], cwd=args.dir, capture_output=True, text=True)
say(f"[e2e] index: exit {r.returncode}, {time.perf_counter() - t:.1f} s")
t = time.perf_counter()
map_exe = exe + "_align" if args.align else exe
r = subprocess.run([map_exe, *common, "-q", "reads.fastq", "-o", "out.sam", *args.extra.split()], cwd=args.dir,
                   capture_output=True, text=True,
                   env={**os.environ, **({"BML_LOG": "1", "BMV_LOG_CLASSES": "1"} if args.verbose else {})})
say(f"[e2e] map ({os.path.basename(map_exe)} {args.extra}): exit {r.returncode}, {time.perf_counter() - t:.1f} s wall")
for line in r.stderr.splitlines():
    if ("[BENCHMARK]" in line or "[ERROR]" in line or line.startswith(("[bm]", "[bmv]", "[bml]")) or "in use at the end" in line
            or "Host peak resident" in line):
        say("    " + line)
truth = [l.split() for l in open(os.path.join(args.dir, "reads.position_ground_truth"))]
names = [g.record_id(i).split(" ")[0] for i in range(g.n_records)]
ok = mapped = 0
seen = set()
for line in open(os.path.join(args.dir, "out.sam")):
    if line[0] == "@":
        continue
    f = line.split("\t", 5)
    i = int(f[0])
    if i in seen:
        continue
    seen.add(i)
    mapped += 1
    ref, pos, rc = int(truth[i][0]), int(truth[i][1]), int(truth[i][2])
    # (long reads: a record's POS is where its first located window starts, up to a read length away)
    ok += int(f[2] == names[ref] and abs(int(f[3]) - pos) <= (sim_len if args.long else 10) and (int(f[1]) == 16) == bool(rc))
say(f"[e2e] out.sam: {os.path.getsize(os.path.join(args.dir, 'out.sam')) / 1e9:.2f} GB")
say(f"[e2e] reads with a SAM record: {mapped}/{rd.n} ({100.0 * mapped / rd.n:.3f} %), first record at the true position "
    f"(+-10): {ok} ({100.0 * ok / rd.n:.3f} %)")
if args.out:
    with open(args.out, "w") as f:
        f.write("\n".join(log) + "\n")
