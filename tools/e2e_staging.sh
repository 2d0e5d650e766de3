#!/bin/bash
# The mapping step of the tools with ordinary and with page-locked read staging (BM_PINNED_STAGING=1), short and long reads.
set -e
mkdir -p gpurun_out
run() {
    name=$1; shift
    python tools/e2e_cli.py "$@" --out gpurun_out/r02_stage_$name.txt > /dev/null 2>&1
    echo "$name: $(grep -E 'map \(' gpurun_out/r02_stage_$name.txt) | $(grep -E 'time for bucket mapping' gpurun_out/r02_stage_$name.txt | cut -c1-90)"
}
run short --workload egu --reads 1000000
BM_PINNED_STAGING=1 run short_pinned --workload egu --reads 1000000
run long --workload grch38 --long --align --bucket-len 262144 --index-seed 10 --reads 20000 --dir /tmp/bm_e2e4
BM_PINNED_STAGING=1 run long_pinned --workload grch38 --long --align --bucket-len 262144 --index-seed 10 --reads 20000 --dir /tmp/bm_e2e4
run short2 --workload egu --reads 1000000
