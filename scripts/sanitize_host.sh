#!/bin/bash
# ASan + UBSan run of the HOST side (SURVEY §5 "Race detection / sanitizers"): builds csrc/build/san/libgraphslam_hip_san.so
# (gs_api.cpp, gs_plan.cpp, gs_slam.cpp, gs_geo.cpp instrumented; device code as usual) and runs the non-GPU tests
# against it through GS_LIB.  CPU box only (GPU sanitizers are not available on this pool).  Exit code = pytest's;
# any sanitizer report aborts the process (halt_on_error) and fails the run.
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC="$ROOT/opendlv-logic-cfsd18-sensation-slam_amd/csrc"
make -C "$CSRC" sanitize > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
export GS_LIB="$CSRC/build/san/libgraphslam_hip_san.so"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_odr_violation=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
cd "$ROOT"
LD_PRELOAD="$RT" python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
