#!/bin/bash
# ThreadSanitizer run of the host threads (SURVEY §5 "Race detection"): the worker pool (tests/tools/pool_tsan.cpp) and the structure phase on it
# (tests/tools/plan_tsan.cpp: gs_plan.cpp compiled with -fsanitize=thread — a single handle, pose-window shards from the window masks and by the
# general recursion, edges in and out of pose order, re-plans on recycled arrays).  CPU box only; exit code 0 = no report.
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
C=opendlv-logic-cfsd18-sensation-slam_amd/csrc
g++ -O1 -g -std=c++17 -fsanitize=thread -I $C tests/tools/pool_tsan.cpp -o /tmp/pool_tsan -lpthread
g++ -O1 -g -std=c++17 -fsanitize=thread -I $C -I include tests/tools/plan_tsan.cpp $C/gs_plan.cpp -o /tmp/plan_tsan -lpthread
export TSAN_OPTIONS=halt_on_error=1:exitcode=66
GS_THREADS=6 /tmp/pool_tsan
GS_THREADS=8 /tmp/plan_tsan 300000
GS_THREADS=3 /tmp/plan_tsan 100000
echo "tsan: clean"
