#!/bin/bash
# PMC passes over the linearisation loop (separate runs per counter group; no tracing domains mixed in)
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/scripts/lin_loop.py cfg4 10 > $OUT.$n.log 2>&1; }
mkdir -p $OUT
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
run c FETCH_SIZE GRBM_GUI_ACTIVE
run d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, os
tag=os.environ.get('TAGX')
for f in sorted(glob.glob('gpurun_out/pmc_*/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'linearize' in k:
            print(f.split('/')[2], k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
find gpurun_out/pmc_$TAG -name "*.csv" -size +2M -delete
