#!/bin/bash
# Experiment helper: bench.py's pruned step under alternative builds of libbmf.so (bucket-map_amd/alt/*.so), e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBMF_RECOUNT_OCC=3 -c -o /tmp/bmf_api_occ3.o bucket-map_amd/csrc/bmf_api.hip
#   hipcc --offload-arch=gfx950 -shared -o bucket-map_amd/alt/libbmf_occ3.so /tmp/bmf_api_occ3.o bucket-map_amd/csrc/bm[lv]_*.o
#   gpurun -- 'bash tools/try_libs.sh'
set -e
shopt -s nullglob
mkdir -p gpurun_out
cp bucket-map_amd/libbmf.so /tmp/libbmf_main.so
for lib in /tmp/libbmf_main.so bucket-map_amd/alt/*.so /tmp/libbmf_main.so; do
    cp "$lib" bucket-map_amd/libbmf.so
    python bench.py --no-pmc --no-extra-legs --steps 10 --warmup 3 > gpurun_out/try_$(basename $lib .so).json
    python - "$lib" <<'PY'
import json, sys, os
d = json.load(open("gpurun_out/try_" + os.path.basename(sys.argv[1])[:-3] + ".json"))
print(os.path.basename(sys.argv[1]), "default", d["ms_per_step"], "pruned", {k: v for k, v in d["pruned"].items() if "ms" in k or "reads" in k})
PY
done
cp /tmp/libbmf_main.so bucket-map_amd/libbmf.so
