#!/bin/bash
# local build (the built .so files travel with the snapshot), then one gpurun call: scripts/gpu.sh TIMEOUT 'command'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/opendlv-logic-cfsd18-sensation-slam_amd/csrc" -j4 > /dev/null
make -C "$ROOT/oracle" > /dev/null
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
