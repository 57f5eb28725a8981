#!/bin/bash
# round 3, first GPU session: parity tests with the g2o-order residuals, parity spreads (new order vs old), baseline bench
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
OLD=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build/var_oldorder/libgraphslam_hip.so
timeout -k 10 300 python scripts/parity_spread.py cfg4 g2o_order > $O/spread_cfg4_new.log 2>&1; echo "spread cfg4 new exit=$?"
GS_LIB=$OLD timeout -k 10 300 python scripts/parity_spread.py cfg4 diff_first > $O/spread_cfg4_old.log 2>&1; echo "spread cfg4 old exit=$?"
timeout -k 10 400 python scripts/parity_spread.py cfg5 g2o_order > $O/spread_cfg5_new.log 2>&1; echo "spread cfg5 new exit=$?"
GS_LIB=$OLD timeout -k 10 400 python scripts/parity_spread.py cfg5 diff_first > $O/spread_cfg5_old.log 2>&1; echo "spread cfg5 old exit=$?"
grep -h "vs\|b_pose" $O/spread_*.log | cut -c1-230
timeout -k 10 600 python bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench exit=$?"
python - <<PY
import json
d = json.load(open("$O/bench_cfg4.json"))
print(round(d["value"]), "it/s", d["phases_ms"], "lin frac", round(d["roofline"]["frac"], 3), "rmse", d["pose_rmse_vs_oracle_rel"])
PY
GS_LIB=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build/var_ts/libgraphslam_hip.so timeout -k 10 200 python scripts/level_times.py cfg4 > $O/level_times_cfg4.txt 2>&1
GS_LIB=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build/var_ts/libgraphslam_hip.so timeout -k 10 300 python scripts/level_times.py cfg5 > $O/level_times_cfg5.txt 2>&1
cat $O/level_times_cfg4.txt
