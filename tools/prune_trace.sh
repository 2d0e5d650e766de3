#!/bin/bash
# Per-kernel times of the pruning kernels under one forced configuration (rocprofv3 --kernel-trace --stats).
#   gpurun -- 'bash tools/prune_trace.sh <tag> <genome-profile> "<config>" [extra prune_probe args]'
set -e
TAG=$1; PROFILE=$2; CFG=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/prune_probe.py --genome-profile "$PROFILE" --configs "$CFG" "$@" > "$OUT/probe.jsonl" 2> "$OUT/probe.err"
cat "$OUT/probe.jsonl"
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms total {float(r["TotalDurationNs"])/1e6:9.2f} ms')
PY
