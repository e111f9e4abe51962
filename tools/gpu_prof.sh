#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats of one bench leg.  bash tools/gpu_prof.sh <tag> <bench args...>
ROOT=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $ROOT/bench.py "$@" > $ROOT/gpurun_out/prof_$TAG.json 2> $ROOT/gpurun_out/prof_$TAG.err
cp /tmp/prof_$TAG/*/*kernel_stats.csv $ROOT/gpurun_out/prof_${TAG}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$ROOT/gpurun_out/prof_${TAG}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('== $TAG total kernel ms', tot/1e6)
for r in rows[:26]:
    print(f"{r['Name'][:96]:96s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
