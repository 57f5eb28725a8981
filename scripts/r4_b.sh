#!/bin/bash
# round 4, experiment B: where the linearisation pass's fetch traffic comes from — FETCH_SIZE / WRITE_SIZE of ablated builds (results wrong, traffic telling)
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
B=$GRAFT_REPO_ROOT/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
cd /tmp && export TMPDIR=/tmp
for v in default abl1 abl2 abl4 abl8 abl15; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    GS_LIB=$L rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$v.$c -- python3 $GRAFT_REPO_ROOT/scripts/lin_loop.py ${1:-cfg4} 10 > $O/$v.$c.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee gpurun_out/r4b/summary.txt
import csv, glob, collections
for v in ("default", "abl1", "abl2", "abl4", "abl8", "abl15"):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("gpurun_out/r4b/%s.%s/*/*counter_collection.csv" % (v, c)):
            for r in csv.DictReader(open(f)):
                if "k_linearize_ell" in r["Kernel_Name"] and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
        out[c] = sum(vals) / max(len(vals), 1)
    print("%-8s launches FETCH_SIZE x2 %9.1f KB  WRITE_SIZE %9.1f KB" % (v, 2 * out["FETCH_SIZE"], out["WRITE_SIZE"]))
PY
for v in default abl1 abl2 abl4 abl8 abl15; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  echo -n "$v: "; GS_LIB=$L python3 scripts/lin_loop.py ${1:-cfg4} 200 | tail -1
done | tee -a gpurun_out/r4b/summary.txt
find gpurun_out/r4b -name "*.csv" -size +1M -delete
