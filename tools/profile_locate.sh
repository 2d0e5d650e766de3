#!/bin/bash
# rocprofv3 over the `bucketmap` tool's map step on the genome-like 1.70 Gbp genome + 1 M reads (the locator's kernels on
# 1.25 G k-mer occurrences): kernel trace + stats, then SQ counters in runs of their own.  gpurun -- 'bash tools/profile_locate.sh r03'
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun}
OUT=$ROOT/gpurun_out/prof_${TAG}_locate
DIR=/tmp/bm_e2e_prof
mkdir -p "$OUT"
python3 "$ROOT/tools/e2e_cli.py" --profile genome --dir $DIR --out "$OUT/e2e.txt" > "$OUT/e2e.log" 2>&1
cd /tmp && export TMPDIR=/tmp && cd $DIR
ARGS="-i idx --genome g.fa --bucket-len 65536 -r 300 -f 1 -q reads.fastq -o out_prof.sam"
rm -f out_prof.sam; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- "$ROOT/bucket-map_amd/bucketmap" $ARGS > "$OUT/kt.log" 2>&1
rm -f out_prof.sam; rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/a" -- "$ROOT/bucket-map_amd/bucketmap" $ARGS > "$OUT/a.log" 2>&1
rm -f out_prof.sam; rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d "$OUT/b" -- "$ROOT/bucket-map_amd/bucketmap" $ARGS > "$OUT/b.log" 2>&1
cd "$ROOT"
python3 tools/pmc_quick.py "$OUT" bml:: > "$OUT/summary.txt"
for f in "$OUT"/kt/*/*_kernel_stats.csv; do grep -E "Name|bml::" "$f" | cut -c1-300 >> "$OUT/summary.txt"; done
cat "$OUT/summary.txt"
