#!/bin/bash
# instruction-mix counters of the solver kernels (separate pmc pass, kernel trace only)
TAG=${1:-pf}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/scripts/iter_loop.py cfg4 3 > $OUT.a.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/scripts/iter_loop.py cfg4 3 > $OUT.b.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_$TAG/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'factor3' in k or 'backsolve3' in k or 'linearize' in k:
            print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
find gpurun_out/pmc_$TAG -name "*.csv" -size +2M -delete
