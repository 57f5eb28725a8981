#!/bin/bash
set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit=$?" | tee -a $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout -k 10 900 python bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench exit=$?"; tail -3 $O/bench_cfg4.err
python - <<PY
import json
d = json.load(open("$O/bench_cfg4.json"))
print(round(d["value"]), "it/s", d["phases_ms"], "lin frac", round(d["roofline"]["frac"], 3), "rmse", d["pose_rmse_vs_oracle_rel"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["repeats"], d["cpu_baseline"]["value_min"], d["cpu_baseline"]["value_max"])
for k, v in d.get("roofline_by_config", {}).items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
for k, v in d.get("wide_view_tracks", {}).items(): print(k, v)
PY
