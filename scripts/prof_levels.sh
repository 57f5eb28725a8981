#!/bin/bash
# per-launch durations (us) of the factor / backsolve level kernels in the last iterations of a short run
TAG=${1:-lv}; CFG=${2:-cfg4}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/scripts/iter_loop.py $CFG 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/prof_$TAG/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last iteration: from the last linearize kernel on
last = max(i for i, n in enumerate(names) if 'linearize_ell' in n)
prev = None
for r in rows[last:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev) / 1e3 if prev else 0.0
    prev = int(r['End_Timestamp'])
    print("%-28s grid %-8s dur %8.1f us  gap %6.1f us" % (r['Kernel_Name'][:28].replace('void gs::',''), r.get('Grid_Size',r.get('Grid_Size_X','?')), d, gap))
PY
rm -rf gpurun_out/prof_$TAG
