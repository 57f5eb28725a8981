#!/bin/bash
# HBM traffic of every kernel of a Gauss-Newton iteration: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (kernel trace only), averaged per kernel over the launches of `reps` iterations.  FETCH_SIZE is printed raw and x2
# (gfx950: a coalesced streaming read reports half its bytes, /opt/skills/guides/MI355X_MICROARCH.md section HBM).
TAG=${1:-it}; CFG=${2:-cfg4}; REPS=${3:-5}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $GRAFT_REPO_ROOT/scripts/iter_loop.py $CFG $REPS > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $GRAFT_REPO_ROOT/scripts/iter_loop.py $CFG $REPS > $OUT/w.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY > gpurun_out/pmc_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('f', 'w'):
    for f in glob.glob('gpurun_out/pmc_$TAG/%s/*/*counter_collection.csv' % tag):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].replace('void gs::', '')[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
print("# $CFG, $REPS iterations; KB per launch (mean over launches); FETCH x2 = gfx950 correction for coalesced streams")
tot = 0.0
for k, v in sorted(acc.items()):
    fs = v.get('FETCH_SIZE', [0]); ws = v.get('WRITE_SIZE', [0])
    fm, wm = sum(fs) / len(fs), sum(ws) / len(ws)
    print("%-42s launches %4d  FETCH_SIZE %10.1f KB (x2 %10.1f)  WRITE_SIZE %10.1f KB" % (k, len(fs), fm, 2 * fm, wm))
PY
cat gpurun_out/pmc_$TAG.txt
find gpurun_out/pmc_$TAG -name "*.csv" -size +1M -delete
