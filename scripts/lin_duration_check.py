"""Reads a rocprofv3 kernel trace (csv) of bench.py and prints the linearisation kernel's mean duration by context: launches that follow
k_update (inside an iteration) and launches that follow another linearisation launch (back to back).  usage: lin_duration_check.py trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev = None; acc = {}
for r in rows:
    name = r["Kernel_Name"]
    if "k_linearize_ell" in name:
        ctx = "after k_update (inside an iteration)" if prev and "k_update" in prev else ("after k_linearize_ell (back to back)" if prev and "k_linearize_ell" in prev else "other")
        acc.setdefault(ctx, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    prev = name
for k, v in acc.items():
    print("k_linearize_ell %-40s launches %4d  mean %.2f us  min %.2f  max %.2f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
