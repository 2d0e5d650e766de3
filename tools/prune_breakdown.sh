#!/bin/bash
# per-kernel times of the pruned step on the genome-like genome for a forced form:  bash tools/prune_breakdown.sh NAME VAR=VALUE...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
export "$@"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pb_$name -- python3 bench.py --no-extra-legs --no-pmc --cpu-sample 0 --genome-profile genome --steps 3 > gpurun_out/pb_$name.json 2> gpurun_out/pb_$name.err
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/pb_$name/*/*_kernel_trace.csv'))[-1]
by=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    if 'bmf::' in k: by[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
print("== $name")
for k,v in sorted(by.items(), key=lambda kv:-sum(kv[1])):
    big=[x for x in v if x>0.5*max(v)]
    if max(v)>0.3: print(f"{k[:60]:60s} n_big={len(big):3d} mean_big={sum(big)/len(big):8.3f} ms")
PY
