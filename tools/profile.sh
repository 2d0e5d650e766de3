#!/bin/bash
# rocprofv3 passes behind the numbers in DESIGN.md / bench.py's `roofline.traffic`.  Run on the GPU box:
#   gpurun -- 'bash tools/profile.sh r01'
# Every pass runs bench.py with its `pruned` leg, so the two-pass pruning kernels are profiled beside the
# headline vote kernel.  Pass 1: kernel trace + stats (per-kernel average duration).  Pass 2 and 3: PMC counters, each in its
# own run (never combined with tracing).  Summaries are written to gpurun_out/prof_<tag>/ and then
# summarised into profiles/<tag>/ by tools/summarize_profile.py (run in the build container).
set -e
TAG=${1:-r02}
WORKLOAD=${2:-egu}
PARAMS=${3:-default}
NAME=${4:-${WORKLOAD}_${PARAMS}}     # 4th argument: a name for the output directory; everything after it goes to bench.py
shift 4 2>/dev/null || shift $#
ROOT=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun (or export it to the repo root)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${NAME}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 5 --warmup 2 --workload "$WORKLOAD" --params "$PARAMS" --cpu-sample 0 --no-pmc --no-extra-legs "$@" > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --workload "$WORKLOAD" --params "$PARAMS" --cpu-sample 0 --no-pmc --no-extra-legs "$@" > "$OUT/bench_pmc.json" 2> "$OUT/bench_pmc.err"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- python3 bench.py --steps 2 --warmup 1 --workload "$WORKLOAD" --params "$PARAMS" --cpu-sample 0 --no-pmc --no-extra-legs "$@" > "$OUT/bench_pmc2.json" 2> "$OUT/bench_pmc2.err"
echo "profiles written under $OUT"
