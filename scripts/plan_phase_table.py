"""Per-build table of the plan phases gs_debug_options.plan_timing prints (stderr of a run): python scripts/plan_phase_table.py FILE"""
import re, sys
runs, cur = [], []
for line in open(sys.argv[1]):
    m = re.match(r"plan phase (\d+): ([\d.]+) ms", line)
    if m:
        if m.group(1) == "0" and cur: runs.append(cur); cur = []
        cur.append((m.group(1), float(m.group(2))))
if cur: runs.append(cur)
print("# phases: 0 index, 1 edge grouping, 21 adjacency, 3 dissection, 4 symbolic, 5 fronts, 70 shard assignment, 71 ELL, 72 records, 73 landmark lists, 74-75 wave tiles")
for i, r in enumerate(runs): print("plan build %d:" % i, " ".join("%s=%.1f" % kv for kv in r), " total %.1f" % sum(v for _, v in r))
