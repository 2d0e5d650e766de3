#!/bin/bash
# One read length under every forced words-per-lane (BMV_CW), to check the cost model of pick_shape against the clock.
#   bash tools/bench_verify_cw.sh LEN READS "2 3 4 5"
set -e
mkdir -p gpurun_out
for cw in $3; do
    BMV_CW=$cw python tools/bench_verify.py --reads $2 --len $1 --indel-rate 0.1 --sub 0.03 --cpu-sample 1 > gpurun_out/vbcw_$1_$cw.json
    python - "$1" "$cw" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/vbcw_{sys.argv[1]}_{sys.argv[2]}.json"))
print("len", sys.argv[1], "BMV_CW", sys.argv[2], f'{d["ms_kernels"]:.2f} ms', f'{d["cell_updates_per_s"] / 1e12:.2f} T cells/s', d["checks"])
PY
done
