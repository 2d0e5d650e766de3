#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 counter_collection CSVs (full-size launches only: largest grid,
and for fixed-grid kernels the long dispatches).  python tools/pmc_quick.py <dir> [kernel substring]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "bmf::"
rows = collections.defaultdict(list)
for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if want in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            rows[(name, r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for (k, c), v in sorted(rows.items()):
    top = max(g for g, _, _ in v)
    v = [(x, t) for g, x, t in v if g == top]
    tmax = max(t for _, t in v)
    v = [x for x, t in v if t > 0.5 * tmax]
    print(f"{k[:50]:50s} {c:22s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
