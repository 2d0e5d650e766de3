#!/bin/bash
# The verifier on a ladder of read lengths (one JSON line each, gpurun_out/vb_<len>.json).
#   bash tools/bench_verify_shapes.sh
set -e
mkdir -p gpurun_out
python tools/bench_verify.py --repeat 3 > gpurun_out/vb_300.json
for spec in "150 1000000" "250 1000000" "500 500000"; do        # one alignment per lane (bmv_align_lane_kernel)
    set -- $spec
    python tools/bench_verify.py --reads $2 --len $1 --cpu-sample 500 --repeat 3 > gpurun_out/vb_$1.json
done
python tools/bench_verify.py --indel-rate 0.1 --sub 0.03 --cpu-sample 500 --repeat 3 > gpurun_out/vb_300noisy.json
for spec in "1000 400000" "2500 200000" "5000 100000" "8000 40000" "10000 40000" "20000 10000"; do
    set -- $spec
    python tools/bench_verify.py --reads $2 --len $1 --indel-rate 0.1 --sub 0.03 --cpu-sample 2 --repeat 2 > gpurun_out/vb_$1.json
done
for f in 150 250 300 300noisy 500 1000 2500 5000 8000 10000 20000; do
    python - "$f" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/vb_{sys.argv[1]}.json"))
print(sys.argv[1], d["config"], f'{d["ms_kernels"]:.2f} ms', f'{d["cell_updates_per_s"] / 1e12:.2f} T cells/s', d["checks"])
PY
done
python tools/bench_verify.py --reads 40000 --len 30000 --mixed 1000 --indel-rate 0.1 --sub 0.03 --cpu-sample 2 --repeat 2 > gpurun_out/vb_mixed.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/vb_mixed.json"))
print("mixed", d["config"], f'{d["ms_kernels"]:.2f} ms', f'{d["cell_updates_per_s"] / 1e12:.2f} T cells/s', d["checks"])
PY
