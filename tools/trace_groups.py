#!/usr/bin/env python3
"""Per-(kernel, grid size) mean durations from a rocprofv3 kernel_trace.csv: tells the tuner's prefix launches from the
full-size ones.   python tools/trace_groups.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
g = defaultdict(list)
for r in csv.DictReader(open(f)):
    g[(r["Kernel_Name"][:70], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for (k, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0.05:
        print(f"{k:70s} grid {grid:>10d} n {len(v):3d} mean {sum(v) / len(v):8.3f} ms  min {min(v):8.3f}")
