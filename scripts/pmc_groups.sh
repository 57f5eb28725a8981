#!/bin/bash
# Counter groups (one rocprofv3 --pmc pass each, kernel trace only) for the kernels of a Gauss-Newton iteration.
# usage: pmc_groups.sh TAG CFG REPS "GROUP1 counters" "GROUP2 counters" ...   -> gpurun_out/pmcg_TAG.txt
TAG=$1; CFG=$2; REPS=$3; shift 3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcg_$TAG; mkdir -p $OUT
i=0
for G in "$@"; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/scripts/iter_loop.py $CFG $REPS > $OUT/g$i.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python3 - <<PY > gpurun_out/pmcg_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcg_$TAG/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].replace('void gs::', '')[:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    if not k.startswith(('k_factor3', 'k_backsolve3', 'k_linearize', 'gs::k_update')): continue
    print(k)
    for c, xs in sorted(v.items()): print("    %-36s %16.1f  (mean of %d launches)" % (c, sum(xs) / len(xs), len(xs)))
PY
cat gpurun_out/pmcg_$TAG.txt
find gpurun_out/pmcg_$TAG -name "*.csv" -size +1M -delete
