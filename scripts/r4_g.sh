#!/bin/bash
# round 4: kernel trace of the launch-bound regime (a lap-sized graph, gs_optimize(10) repeated): per-kernel durations and the gaps between them
cd /tmp && export TMPDIR=/tmp
for sz in "240 200" "1000 200"; do
  tag=$(echo $sz | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small_$tag -- python3 $GRAFT_REPO_ROOT/scripts/small_trace.py $sz 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_small_$tag.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_small_$tag.log
  python3 - <<PY
import csv, glob
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_small_$tag/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# one gs_optimize(10) call in the middle: find update kernels, take iterations between the 50th and 60th k_update
upd = [i for i, r in enumerate(rows) if 'k_update' in r['Kernel_Name']]
a, b = upd[50], upd[60]
prev = None; acc = {}
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].replace('void gs::', '').replace('gs::', '')[:34]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev) / 1e3 if prev else 0.0
    prev = int(r['End_Timestamp'])
    e = acc.setdefault(n, [0, 0.0, 0.0]); e[0] += 1; e[1] += d; e[2] += gap
tot = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3
print("  10 iterations: %.1f us wall on the device" % tot)
for n, e in acc.items(): print("  %-36s launches %3d  mean duration %6.2f us  mean gap before %5.2f us" % (n, e[0], e[1] / e[0], e[2] / e[0]))
PY
done
