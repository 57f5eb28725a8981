#!/bin/bash
# round 3, last check of the committed state: GPU suite, smoke, the default bench line (as the driver runs it)
O=gpurun_out/r3x; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit=$rc"; tail -4 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit=$?"; tail -1 $O/smoke.log
t0=$(date +%s); timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit=$? in $(( $(date +%s) - t0 )) s"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_default.json") if l.startswith('{"metric')][-1])
r = d["roofline"]; print("bench: %.0f it/s, %.4f ms/step; roofline frac %.3f traffic %s (%s); solver traffic %s; growth %s" % (d["value"], d["ms_per_step"], r["frac"], r["traffic"], r["traffic_source"], d["solver"]["traffic"], d.get("growth")))
PY
