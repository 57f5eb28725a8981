#!/bin/bash
# round 3: the linearisation kernel's duration from events attached to its dispatch vs the rocprofv3 kernel trace of the same command
O=gpurun_out/r3t; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --no-cpu --no-extra-configs --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
python - <<PY
import json
t = [l for l in open("$O/bench.json") if l.startswith('{"metric')][-1]; d = json.loads(t)
r = d["roofline"]; print("bench: %.0f it/s; roofline ms_per_launch %.5f (event to event %.5f) frac %.3f; back to back %.5f" % (d["value"], r["ms_per_launch"], r["ms_event_to_event"], r["frac"], r["ms_per_launch_back_to_back"]))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --no-cpu --no-extra-configs --steps 100 --warmup 10 > $R/$O/prof.log 2>&1; echo "rocprof exit=$?"
cd $R
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/lin_duration_check.py "$f" | tee $O/lin_duration_check.txt
python - <<PY
import json
t = [l for l in open("$O/prof.log") if l.startswith('{"metric')][-1]; d = json.loads(t)
r = d["roofline"]; print("under the profiler: roofline ms_per_launch %.5f (event to event %.5f)" % (r["ms_per_launch"], r["ms_event_to_event"]))
PY
