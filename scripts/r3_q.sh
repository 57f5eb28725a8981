#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in "50 30" "240 200" "1000 200"; do timeout -k 10 120 python3 $R/scripts/small_trace.py $s 2>&1 | tee -a $R/$O/small.txt; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o small -- python3 $R/scripts/small_trace.py 240 200 > $R/$O/prof.log 2>&1; echo "rocprof exit=$?"
find $R/$O/prof -name "*kernel_stats*" | head -3
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" | cut -c1-200
