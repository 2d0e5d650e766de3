#!/bin/bash
# SQ counters of the bench's kernels, two passes (8 SQ slots each).  gpurun -- 'bash tools/pmc_sq.sh <tag> [bench args]'
set -e
TAG=${1:-sq}
shift || true
ROOT=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/a" -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-pmc --no-extra-legs "$@" > "$OUT/a.json" 2> "$OUT/a.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/b" -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-pmc --no-extra-legs "$@" > "$OUT/b.json" 2> "$OUT/b.err"
python3 tools/pmc_quick.py "$OUT" bmf:: > "$OUT/summary.txt"
cat "$OUT/summary.txt"
