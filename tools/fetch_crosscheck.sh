#!/bin/bash
# Cross-check of the x2 that MI355X_MICROARCH.md prescribes for FETCH_SIZE on gfx950 (wide streaming reads), against the
# L2's own request-size counters: bytes = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B, and against
# TCC_EA0_RDREQ_DRAM_32B x 32 B (requests ADDRESSED to DRAM, Infinity-Cache hits included: no counter sits behind that
# cache).  Vote kernel of bench.py's headline workload; two passes of counters.
#   gpurun -- 'bash tools/fetch_crosscheck.sh'   ->  gpurun_out/fetch_crosscheck/summary.txt
set -e
ROOT=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}
OUT=$ROOT/gpurun_out/fetch_crosscheck
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
ARGS="--steps 2 --warmup 1 --cpu-sample 0 --no-pmc --no-extra-legs --no-pruned-leg"
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d "$OUT/a" -- python3 bench.py $ARGS > "$OUT/a.json" 2> "$OUT/a.err"
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_DRAM_sum --output-format csv -d "$OUT/b" -- python3 bench.py $ARGS > "$OUT/b.json" 2> "$OUT/b.err" || true
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
algo = json.loads(open(os.path.join(out, "a.json")).read().strip().splitlines()[-1])["roofline"]["algorithmic_bytes_per_launch"]
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "bmf_vote_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: max(v) for k, v in acc.items()}       # the full-size dispatches (the warm-up pieces are smaller)
print(f"vote kernel, full-size dispatch; algorithmic bytes per launch {algo:.4e}")
for k in sorted(m):
    print(f"  {k:28s} {m[k]:.6e}  (n={len(acc[k])})")
if "FETCH_SIZE" in m:
    print(f"FETCH_SIZE x 1024 x 2            = {m['FETCH_SIZE'] * 2048:.4e} B = {m['FETCH_SIZE'] * 2048 / algo:.4f} x algorithmic")
if "TCC_EA0_RDREQ_sum" in m:
    r, r32 = m["TCC_EA0_RDREQ_sum"], m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    print(f"RDREQ: {r:.4e} requests, {r32:.3e} of them 32 B; if the others are 128 B: {(r - r32) * 128 + r32 * 32:.4e} B = {((r - r32) * 128 + r32 * 32) / algo:.4f} x algorithmic; "
          f"if 64 B: {((r - r32) * 64 + r32 * 32) / algo:.4f} x")
if "TCC_EA0_RDREQ_128B" in m or "TCC_EA0_RDREQ_64B" in m:
    b = m.get("TCC_EA0_RDREQ_128B", 0) * 128 + m.get("TCC_EA0_RDREQ_64B", 0) * 64
    print(f"64 B x RDREQ_64B + 128 B x RDREQ_128B = {b:.4e} B = {b / algo:.4f} x algorithmic")
if "TCC_EA0_RDREQ_DRAM_32B" in m:
    print(f"RDREQ_DRAM_32B x 32 B            = {m['TCC_EA0_RDREQ_DRAM_32B'] * 32:.4e} B = {m['TCC_EA0_RDREQ_DRAM_32B'] * 32 / algo:.4f} x algorithmic")
PY
