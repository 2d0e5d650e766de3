#!/usr/bin/env python3
"""Summarises the rocprofv3 CSVs of tools/profile.sh into profiles/<tag>/ (tracked) and refreshes
profiles/pmc_latest.json (what bench.py quotes when it cannot run the profiler itself).

    python tools/summarize_profile.py gpurun_out/prof_r02_egu_default r02 egu default 1000000

bench.py launches every kernel at two sizes -- the timed loop's full batch and the 64 Ki-window pieces of the
PCIe-inclusive leg -- so everything here is restricted to a kernel's LARGEST grid: those are the launches
`roofline` prices.  Per-dispatch durations and counter values are kept, not only their means.
"""
import collections
import csv
import glob
import json
import os
import sys

src, tag, workload, params, reads = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
name = os.path.basename(os.path.normpath(src)).split(tag + "_", 1)[-1]   # prof_<tag>_<name>
stem = "bench_" + name
OURS = ("bmf::", "bmi::", "bml::", "bmv::")


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


# ---- kernel trace: per-dispatch durations of the full-size launches
by_kernel = collections.defaultdict(list)
for r in csv.DictReader(open(newest(os.path.join(src, "kt", "*", "*_kernel_trace.csv")))):
    if any(t in r["Kernel_Name"] for t in OURS):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        by_kernel[short(r["Kernel_Name"])].append((grid, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
with open(os.path.join(dst, stem + "_kernel_stats.csv"), "w") as f:
    f.write("kernel,full_size_grid_threads,dispatches,mean_ms,min_ms,max_ms,all_dispatches_of_any_size,per_dispatch_ms\n")
    for k, v in sorted(by_kernel.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        top = max(g for g, _ in v)
        d = [ms for g, ms in v if g == top]
        d = [ms for ms in d if ms > 0.5 * max(d)]      # fixed-grid kernels: the full batch's launches are the long ones
        f.write(f"\"{k}\",{top},{len(d)},{sum(d) / len(d):.4f},{min(d):.4f},{max(d):.4f},{len(v)},\"{' '.join(f'{x:.3f}' for x in d)}\"\n")

# ---- counters: per-dispatch values of the full-size launches
summary = {}
for d in ("pmc_fetch", "pmc_l2"):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(os.path.join(src, d, "*", "*_counter_collection.csv")))):
        if any(t in r["Kernel_Name"] for t in OURS):
            rows[(short(r["Kernel_Name"]), r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for (k, c), v in sorted(rows.items()):
        top = max(g for g, _ in v)
        vals = [x for g, x in v if g == top]
        vals = [x for x in vals if x > 0.5 * max(vals)] if max(vals) > 0 else vals
        summary.setdefault(k, {})[c] = {"grid_threads": top, "dispatches": len(vals), "mean": sum(vals) / len(vals), "per_dispatch": vals}
json.dump(summary, open(os.path.join(dst, stem + "_pmc_summary.json"), "w"), indent=1)

for fname in ("bench_kt.json", "bench_pmc.json"):      # the bench lines of the profiled runs themselves
    p = os.path.join(src, fname)
    if os.path.exists(p) and open(p).read().strip():
        line = open(p).read().strip().splitlines()[-1]
        open(os.path.join(dst, f"{stem}_{fname.replace('bench_', '')}"), "w").write(line + "\n")

# the headline kernel: the vote kernel without pruning (template arguments ..., false, false>)
vote = next(k for k in summary if "bmf_vote_kernel" in k and k.rstrip().endswith("false>"))   # PRUNE = false
fetch_kib = summary[vote]["FETCH_SIZE"]["mean"]
latest = {
    "workload": workload, "params": params, "reads": reads, "kernel": vote,
    "FETCH_SIZE_KiB_per_launch": fetch_kib,
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KiB and reads exactly 1/2 of a wide streaming read on gfx950
    "vote_kernel_traffic_bytes": int(fetch_kib * 1024 * 2),
    "l2_hit_rate": summary[vote]["TCC_HIT_sum"]["mean"] / (summary[vote]["TCC_HIT_sum"]["mean"] + summary[vote]["TCC_MISS_sum"]["mean"]),
    "source": f"profiles/{tag}/{stem}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE, own pass; x1024 x2)",
}
# Bytes the pruning kernels fetched.  Pass 1 streams whole rows like the vote kernel (x2 applies); the recount
# kernel issues lone 16-byte loads, one 64-byte sector each, which FETCH_SIZE tallies at face value (checked against
# the kernel's own load count): no x2 there.
for k in summary:
    if "FETCH_SIZE" not in summary[k]:
        continue
    kib = summary[k]["FETCH_SIZE"]["mean"]
    if "bmf_pass1_kernel" in k:
        latest.setdefault("pruning_kernels_traffic_bytes", {})[k] = int(kib * 1024 * 2)
    elif "bmf_recount_kernel" in k or "bmf_vote2_slow_kernel" in k:
        latest.setdefault("pruning_kernels_traffic_bytes", {})[k] = int(kib * 1024)
print(json.dumps(latest, indent=1))
if name != f"{workload}_{params}":      # a one-off geometry (extra bench flags): not what bench.py's default run is
    sys.exit(0)
path = os.path.join(root, "profiles", "pmc_latest.json")
try:
    table = json.load(open(path))
    if "entries" not in table:
        table = {"entries": [table]}
except (OSError, ValueError):
    table = {"entries": []}
table["entries"] = [e for e in table["entries"] if (e.get("workload"), e.get("params")) != (workload, params)] + [latest]
json.dump(table, open(path, "w"), indent=1)
