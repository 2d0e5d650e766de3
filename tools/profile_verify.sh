#!/bin/bash
# rocprofv3 passes over the alignment verifier on 40 000 alignments of 10 000 x 11 001: kernel trace + stats, then SQ
# counters (own runs, never combined with tracing).  gpurun -- 'bash tools/profile_verify.sh r02'
set -e
TAG=${1:-r02}
NAME=${2:-verify}                     # 2nd argument: a name for the output directory; everything after it replaces ARGS
ROOT=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun}
OUT=$ROOT/gpurun_out/prof_${TAG}_${NAME}
mkdir -p "$OUT"
ARGS="--reads 40000 --len 10000 --indel-rate 0.1 --sub 0.03 --cpu-sample 1 --repeat 2"
if [ $# -gt 2 ]; then shift 2; ARGS="$*"; fi
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 tools/bench_verify.py $ARGS > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/a" -- python3 tools/bench_verify.py $ARGS > "$OUT/a.json" 2> "$OUT/a.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/b" -- python3 tools/bench_verify.py $ARGS > "$OUT/b.json" 2> "$OUT/b.err"
python3 tools/pmc_quick.py "$OUT" bmv:: > "$OUT/summary.txt"
for f in "$OUT"/kt/*/*_kernel_stats.csv; do grep -E "Name|bmv::" "$f" | cut -c1-260 >> "$OUT/summary.txt"; done
cat "$OUT/summary.txt"
