#!/bin/bash
# calibrate FETCH_SIZE / WRITE_SIZE on known 1 GiB streams (separate pmc passes), then the linearisation kernel
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- $GRAFT_REPO_ROOT/scripts/calib/fetch_calib > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- $GRAFT_REPO_ROOT/scripts/calib/fetch_calib > $OUT/w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/lf -- python3 $GRAFT_REPO_ROOT/scripts/lin_loop.py cfg4 20 > $OUT/lf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/lw -- python3 $GRAFT_REPO_ROOT/scripts/lin_loop.py cfg4 20 > $OUT/lw.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for tag in ('f', 'w', 'lf', 'lw'):
    for f in glob.glob('gpurun_out/pmc_calib/%s/*/*counter_collection.csv' % tag):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:34], r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            if any(s in k for s in ('read', 'write', 'linearize_ell')):
                print("%-3s %-36s %-11s mean %.1f KB  (n=%d)" % (tag, k, c, sum(v) / len(v), len(v)))
PY
find gpurun_out/pmc_calib -name "*.csv" -size +1M -delete
