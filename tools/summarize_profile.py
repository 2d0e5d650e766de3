#!/usr/bin/env python3
"""Summarises the rocprofv3 CSVs of tools/profile.sh into profiles/<tag>/ (tracked) and refreshes
profiles/pmc_latest.json, which bench.py reads for `roofline.traffic`.

    python tools/summarize_profile.py gpurun_out/prof_r01_egu_default r01 egu default 1000000
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, workload, params, reads = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
stem = f"bench_{workload}_{params}"


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's."""
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv")), os.path.join(dst, stem + "_kernel_stats.csv"))
summary = {}
for d in ("pmc_fetch", "pmc_l2"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(os.path.join(src, d, "*", "*_counter_collection.csv")))):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        summary.setdefault(k, {})[c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
json.dump(summary, open(os.path.join(dst, stem + "_pmc_summary.json"), "w"), indent=1)
# the headline kernel: the vote kernel without pruning (template arguments ..., false, false>)
vote = next(k for k in summary if "bmf_vote_kernel" in k and k.rstrip().endswith("false, false>"))
fetch_kib = summary[vote]["FETCH_SIZE"]["mean"]
latest = {
    "workload": workload, "params": params, "reads": reads, "kernel": vote,
    "FETCH_SIZE_KiB_per_launch": fetch_kib,
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KiB and reads exactly 1/2 of a wide streaming read on gfx950
    "vote_kernel_traffic_bytes": int(fetch_kib * 1024 * 2),
    "l2_hit_rate": summary[vote]["TCC_HIT_sum"]["mean"] / (summary[vote]["TCC_HIT_sum"]["mean"] + summary[vote]["TCC_MISS_sum"]["mean"]),
    "source": f"profiles/{tag}/{stem}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE, own pass; x1024 x2)",
}
# bytes the pruning kernels really fetched, for DESIGN.md 4.2 (same x1024 x2 correction)
for k in summary:
    if "FETCH_SIZE" in summary[k] and any(n in k for n in ("bmf_pass1_kernel", "bmf_recount_kernel", "bmf_vote2_slow_kernel")):
        latest.setdefault("pruning_kernels_traffic_bytes", {})[k] = int(summary[k]["FETCH_SIZE"]["mean"] * 1024 * 2)
# one entry per profiled (workload, params): bench.py picks the one that matches its run
path = os.path.join(root, "profiles", "pmc_latest.json")
try:
    table = json.load(open(path))
    if "entries" not in table:
        table = {"entries": [table]}
except (OSError, ValueError):
    table = {"entries": []}
table["entries"] = [e for e in table["entries"] if (e.get("workload"), e.get("params")) != (workload, params)] + [latest]
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(latest, indent=1))
