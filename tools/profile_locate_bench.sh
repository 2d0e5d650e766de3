#!/bin/bash
# rocprofv3 over tools/bench_locate.py (bench.py's locator leg): kernel trace + stats, then SQ counters in passes of their own.
#   gpurun -- 'bash tools/profile_locate_bench.sh r04 genome'      (second argument: uniform | genome)
set -e
TAG=${1:-r04}
PROFILE=${2:-genome}
ROOT=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun}
OUT=$ROOT/gpurun_out/prof_${TAG}_locate_${PROFILE}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
CMD="python3 tools/bench_locate.py --genome-profile $PROFILE --calls 2"
$CMD > "$OUT/plain.json" 2> "$OUT/plain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $CMD > "$OUT/kt.json" 2> "$OUT/kt.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/a" -- $CMD > "$OUT/a.json" 2> "$OUT/a.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d "$OUT/b" -- $CMD > "$OUT/b.json" 2> "$OUT/b.err"
python3 tools/pmc_quick.py "$OUT" bml:: > "$OUT/summary.txt"
for f in "$OUT"/kt/*/*_kernel_stats.csv; do grep -E "Name|bml::" "$f" | cut -c1-400 >> "$OUT/summary.txt"; done
echo "--- plain run (no profiler)" >> "$OUT/summary.txt"
cat "$OUT/plain.json" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
