#!/bin/bash
# round 4, batch J: GPU suite (with the 100k-pose wide-view parity cases), then the structure phase: single handle and rank handles of 8 x cfg4
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > $O/gpu_suite.txt 2>&1; rc=$?; grep -E "^K=|passed|failed|error" $O/gpu_suite.txt | tail -12
[ $rc -ne 0 ] && exit $rc
GS_PLAN_TIMING=1 timeout -k 10 600 python scripts/shard_footprint.py 8 cfg4 > $O/shard_footprint.txt 2> $O/shard_plan_phases.txt; cat $O/shard_footprint.txt
python - <<'PY'
import re
runs, cur = [], []
for line in open("gpurun_out/r4j/shard_plan_phases.txt"):
    m = re.match(r"plan phase (\d+): ([\d.]+) ms", line)
    if m:
        if m.group(1) == "0" and cur: runs.append(cur); cur = []
        cur.append((m.group(1), float(m.group(2))))
if cur: runs.append(cur)
for i, r in enumerate(runs): print("plan build %d:" % i, " ".join("%s=%.1f" % kv for kv in r), " total %.1f" % sum(v for _, v in r))
PY
