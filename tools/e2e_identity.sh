#!/bin/bash
# The tools on the 1 M-read workload with one device context and with three (the multi-GPU split run on one card):
# the SAM files must be byte-identical.  Writes gpurun_out/e2e_identity/<name>.{txt,md5}.
#   bash tools/e2e_identity.sh [extra e2e_cli.py flags, e.g. --profile genome]
set -e
mkdir -p gpurun_out/e2e_identity
EXTRA="$@"
run() {
    name=$1; shift
    python tools/e2e_cli.py --workload egu --reads 1000000 $EXTRA "$@" --out gpurun_out/e2e_identity/$name.txt > gpurun_out/e2e_identity/$name.log 2>&1
    md5sum /tmp/bm_e2e/out.sam | cut -c1-32 > gpurun_out/e2e_identity/$name.md5
    echo "$name $(cat gpurun_out/e2e_identity/$name.md5)"
}
run one
run three --gpus 0,0,0
run one_align --align
run three_align --align --gpus 0,0,0
cmp gpurun_out/e2e_identity/one.md5 gpurun_out/e2e_identity/three.md5
cmp gpurun_out/e2e_identity/one_align.md5 gpurun_out/e2e_identity/three_align.md5
echo identical
