#!/bin/bash
# round 3, final: the GPU suite, then every measurement the documents cite (scripts/profile_round_r3.sh)
O=gpurun_out/r3v; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit=$rc"; tail -4 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit=$?"; tail -2 $O/smoke.log
bash scripts/profile_round_r3.sh
