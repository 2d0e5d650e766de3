set -e
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --no-extra-legs --no-pmc --cpu-sample 0 --genome-profile genome --steps 3 > gpurun_out/pv_$name.json 2> gpurun_out/pv_$name.err || echo "$name failed"; python3 - <<PY
import json
d=json.load(open("gpurun_out/pv_$name.json"))
p=d["pruned"]
print("$name", round(p["ms_per_step"],2), p.get("form"), p.get("pass1_rows"), p.get("pass1_fold"), p.get("pass1_fold_rows"), p.get("items_recounted"), p.get("items_slow_path"), p.get("recount_column_loads"), p["outputs_identical_to_headline_run"])
PY
}
run auto BMF_X=1
run f4r3 BMF_FOLD=4 BMF_FOLD_ROWS=3
run f4r4 BMF_FOLD=4 BMF_FOLD_ROWS=4
run f2r2_32 BMF_FOLD=2 BMF_FOLD_ROWS=2 BMF_MAX_LIVE=32
run f2r3_16 BMF_FOLD=2 BMF_FOLD_ROWS=3 BMF_MAX_LIVE=16
run u1_32 BMF_FOLD=0 BMF_PASS1_ROWS=1 BMF_MAX_LIVE=32
run u2_32 BMF_FOLD=0 BMF_PASS1_ROWS=2 BMF_MAX_LIVE=32
